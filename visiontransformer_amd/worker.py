"""`predict()` worker service: the "external model server" the reference's backend talks to but
does not contain.

Wire contract, taken from the caller (backend/core/views.py:97-149, backend/project/settings.py:186-187):

* inbound   POST ORCH_URL (default http://127.0.0.1:8001/enqueue/), multipart/form-data with the file `input_image`
            and the fields `job_id`, `vision_model_id`; header `X-ORCH-TOKEN: <ORCH_SHARED_TOKEN>`.  The caller waits at
            most 60 s and only checks for status **202** -- so the image is queued and 202 is returned at once.
* outbound  POST <backend>/api/inference-jobs/<job_id>/complete/ , multipart with the file `mask_image`; the backend
            stores it and flips the job to DONE (a second completion of the same job is answered 400 and ignored here).

Between the two, a single GPU thread drains the queue: all waiting jobs of the same model are decoded, pre-processed on
the device (preprocess.Preprocessor) and pushed through ONE `predict_mask` call (cross-job batching keeps the MI355X
fed; a ViT-B/16 forward at batch 1 and at batch 8 cost nearly the same), then each mask is colourised
(`index_to_color[mask]`, testViTModel.py:139-143), PNG-encoded and posted back.

    python -m visiontransformer_amd.worker --port 8001 --backend http://127.0.0.1:8000 \\
           --model 0:17:/ckpt/version_0.ckpt          # vision_model_id:num_classes[:checkpoint]
"""
from __future__ import annotations

import argparse
import io
import json
import logging
import os
import queue
import threading
import time
import urllib.error
import urllib.request
import uuid
from dataclasses import dataclass, field
from email.parser import BytesParser
from email.policy import HTTP
from http.server import BaseHTTPRequestHandler, ThreadingHTTPServer
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np

log = logging.getLogger("vitseg.worker")


def default_palette(num_classes: int) -> np.ndarray:
    """uint8 [C, 3] colours for class indices (class 0 black), deterministic."""
    pal = np.zeros((max(num_classes, 1), 3), np.uint8)
    for c in range(1, num_classes):
        pal[c] = [(c * 67) % 256, (c * 131 + 80) % 256, (c * 197 + 160) % 256]
    return pal


def parse_multipart(content_type: str, body: bytes):
    """(fields: name -> str, files: name -> (filename, bytes)) of a multipart/form-data body."""
    msg = BytesParser(policy=HTTP).parsebytes(b"Content-Type: " + content_type.encode() + b"\r\n\r\n" + body)
    if not msg.is_multipart():
        raise ValueError("multipart/form-data expected")
    fields, files = {}, {}
    for part in msg.iter_parts():
        name = part.get_param("name", header="content-disposition")
        if name is None:
            continue
        payload = part.get_payload(decode=True) or b""
        fn = part.get_filename()
        if fn is not None:
            files[name] = (fn, payload)
        else:
            fields[name] = payload.decode("utf-8", "replace")
    return fields, files


def encode_multipart(files: Dict[str, tuple], fields: Optional[Dict[str, str]] = None):
    """(content_type, body) for `requests.post(url, data=fields, files=files)`-style uploads, stdlib only."""
    boundary = "----vitseg" + uuid.uuid4().hex
    out = io.BytesIO()
    for k, v in (fields or {}).items():
        out.write(f'--{boundary}\r\nContent-Disposition: form-data; name="{k}"\r\n\r\n{v}\r\n'.encode())
    for k, (fn, data, ctype) in files.items():
        out.write(f'--{boundary}\r\nContent-Disposition: form-data; name="{k}"; filename="{fn}"\r\n'
                  f"Content-Type: {ctype}\r\n\r\n".encode())
        out.write(data)
        out.write(b"\r\n")
    out.write(f"--{boundary}--\r\n".encode())
    return f"multipart/form-data; boundary={boundary}", out.getvalue()


def png_bytes(rgb: np.ndarray) -> bytes:
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(rgb, "RGB").save(buf, format="PNG")
    return buf.getvalue()


@dataclass
class Job:
    job_id: str
    vision_model_id: int
    image: bytes
    received: float = field(default_factory=time.time)


class ModelSlot:
    """One `VisionModel` row of the backend (backend/core/models.py:29-33: num_classes, input_size) bound to a loaded
    network.  `predict_batch(list of uint8 HWC arrays) -> list of uint8 [S, S] masks`."""

    def __init__(self, predict_batch: Callable[[Sequence[np.ndarray]], List[np.ndarray]], num_classes: int,
                 palette: Optional[np.ndarray] = None):
        self.predict_batch = predict_batch
        self.num_classes = num_classes
        self.palette = default_palette(num_classes) if palette is None else np.asarray(palette, np.uint8)


def gpu_slot(model_id_or_config, num_classes: int, checkpoint: Optional[str] = None, *, image_size: int = 224,
             precision: str = "fp32", device: str = "cuda:0") -> ModelSlot:
    """ModelSlot running on libvitseg: device pre-processing of every image (sizes may differ per job), one batched
    forward + fused sigmoid/argmax."""
    import torch
    from .predict import load_model
    from .preprocess import Preprocessor
    model = load_model(model_id_or_config, num_classes, checkpoint, image_size=image_size, precision=precision,
                       device=device)
    seg, pre = model.model, Preprocessor(image_size, device)

    def predict_batch(images):
        x = torch.empty((len(images), 3, image_size, image_size), dtype=torch.float32, device=device)
        for i, a in enumerate(images):
            pre.images(torch.from_numpy(np.ascontiguousarray(a)), out=x[i:i + 1])
        with torch.no_grad():
            m = seg.predict_mask(x)
        return list(m.cpu().numpy())

    slot = ModelSlot(predict_batch, num_classes)
    slot.model = model   # the LightningViTModel, e.g. to load weights after construction
    return slot


class Worker:
    """Queue + GPU thread + completion callbacks.  `backend_url` is the Django base URL; `post` can be replaced in tests."""

    def __init__(self, slots: Dict[int, ModelSlot], backend_url: str, token: str, max_batch: int = 16,
                 batch_wait_s: float = 0.005):
        self.slots, self.backend_url, self.token = slots, backend_url.rstrip("/"), token
        self.max_batch, self.batch_wait_s = max_batch, batch_wait_s
        self.q: "queue.Queue[Job]" = queue.Queue()
        self.stats = dict(received=0, completed=0, failed=0, batches=0, largest_batch=0)
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, name="vitseg-gpu", daemon=True)

    def start(self):
        self._thread.start()
        return self

    def stop(self):
        self._stop.set()
        self._thread.join(timeout=10)

    def submit(self, job: Job):
        self.stats["received"] += 1
        self.q.put(job)

    # ---- GPU thread -------------------------------------------------------------------------------------------
    def _take_batch(self) -> List[Job]:
        try:
            first = self.q.get(timeout=0.1)
        except queue.Empty:
            return []
        batch, later = [first], []
        deadline = time.time() + self.batch_wait_s     # a short wait lets concurrent uploads share one forward
        while len(batch) < self.max_batch:
            try:
                j = self.q.get(timeout=max(0.0, deadline - time.time()))
            except queue.Empty:
                break
            (batch if j.vision_model_id == first.vision_model_id else later).append(j)
        for j in later:
            self.q.put(j)
        return batch

    def _run(self):
        from .predict import decode
        while not self._stop.is_set():
            batch = self._take_batch()
            if not batch:
                continue
            slot = self.slots.get(batch[0].vision_model_id)
            try:
                if slot is None:
                    raise KeyError(f"unknown vision_model_id {batch[0].vision_model_id}")
                masks = slot.predict_batch([decode(j.image) for j in batch])
                self.stats["batches"] += 1
                self.stats["largest_batch"] = max(self.stats["largest_batch"], len(batch))
                for j, m in zip(batch, masks):
                    self._complete(j, png_bytes(slot.palette[np.minimum(m, len(slot.palette) - 1)]))
            except Exception:
                log.exception("batch of %d jobs failed", len(batch))
                self.stats["failed"] += len(batch)

    def post(self, url: str, content_type: str, body: bytes) -> int:
        req = urllib.request.Request(url, data=body, method="POST", headers={"Content-Type": content_type})
        try:
            with urllib.request.urlopen(req, timeout=60) as r:
                return r.status
        except urllib.error.HTTPError as e:
            return e.code

    def _complete(self, job: Job, png: bytes):
        ctype, body = encode_multipart({"mask_image": (f"{job.job_id}_mask.png", png, "image/png")})
        code = self.post(f"{self.backend_url}/api/inference-jobs/{job.job_id}/complete/", ctype, body)
        if code == 200:
            self.stats["completed"] += 1
        else:   # 400 = job already DONE (views.py:128-132): nothing to retry
            log.warning("backend answered %s for job %s", code, job.job_id)
            self.stats["failed"] += 1

    # ---- HTTP front --------------------------------------------------------------------------------------------
    def handler(self):
        worker = self

        class Handler(BaseHTTPRequestHandler):
            def log_message(self, fmt, *args):
                log.debug(fmt, *args)

            def _reply(self, code, obj):
                data = json.dumps(obj).encode()
                self.send_response(code)
                self.send_header("Content-Type", "application/json")
                self.send_header("Content-Length", str(len(data)))
                self.end_headers()
                self.wfile.write(data)

            def do_GET(self):
                if self.path.rstrip("/") == "/health":
                    return self._reply(200, dict(status="ok", queued=worker.q.qsize(), **worker.stats))
                self._reply(404, dict(error="not found"))

            def do_POST(self):
                if self.path.rstrip("/") != "/enqueue":
                    return self._reply(404, dict(error="not found"))
                if self.headers.get("X-ORCH-TOKEN") != worker.token:
                    return self._reply(403, dict(error="bad X-ORCH-TOKEN"))
                try:
                    n = int(self.headers.get("Content-Length", "0"))
                    fields, files = parse_multipart(self.headers.get("Content-Type", ""), self.rfile.read(n))
                    job = Job(fields["job_id"], int(fields["vision_model_id"]), files["input_image"][1])
                except (KeyError, ValueError) as e:
                    return self._reply(400, dict(error=f"input_image, job_id and vision_model_id are required ({e})"))
                if job.vision_model_id not in worker.slots:
                    return self._reply(404, dict(error=f"vision_model_id {job.vision_model_id} is not loaded"))
                worker.submit(job)
                self._reply(202, dict(job_id=job.job_id, queued=worker.q.qsize()))

        return Handler

    def serve(self, host: str = "127.0.0.1", port: int = 8001) -> ThreadingHTTPServer:
        """Starts the GPU thread and returns a bound (not yet serving) HTTP server; call `.serve_forever()`."""
        self.start()
        return ThreadingHTTPServer((host, port), self.handler())


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--host", default="127.0.0.1")
    ap.add_argument("--port", type=int, default=8001)
    ap.add_argument("--backend", default=os.environ.get("BACKEND_URL", "http://127.0.0.1:8000"))
    ap.add_argument("--token", default=os.environ.get("ORCH_SHARED_TOKEN", "your_shared_secret_token"))
    ap.add_argument("--model", action="append", required=True,
                    help="vision_model_id:num_classes[:checkpoint[:config_id[:image_size]]] (repeatable)")
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--max-batch", type=int, default=16)
    a = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO)
    slots = {}
    for spec in a.model:
        parts = spec.split(":")
        mid, C = int(parts[0]), int(parts[1])
        ck = parts[2] if len(parts) > 2 and parts[2] else None
        cfg_id = int(parts[3]) if len(parts) > 3 else 0
        size = int(parts[4]) if len(parts) > 4 else 224
        slots[mid] = gpu_slot(cfg_id, C, ck, image_size=size, precision=a.precision)
    srv = Worker(slots, a.backend, a.token, max_batch=a.max_batch).serve(a.host, a.port)
    log.info("listening on %s:%d/enqueue/ -> %s", a.host, a.port, a.backend)
    srv.serve_forever()


if __name__ == "__main__":
    main()
