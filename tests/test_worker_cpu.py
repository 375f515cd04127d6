"""CPU: the predict() worker's host logic (row f2) -- the orchestrator wire contract of backend/core/views.py:97-149
exercised over real sockets with a stand-in backend and a stand-in network (no GPU)."""
import io
import json
import threading
import time
import urllib.error
import urllib.request
from http.server import BaseHTTPRequestHandler, ThreadingHTTPServer

import numpy as np
import pytest

from visiontransformer_amd.worker import ModelSlot, Worker, default_palette, encode_multipart, parse_multipart

Image = pytest.importorskip("PIL.Image")
TOKEN = "s3cret"


class FakeBackend:
    """Plays Django: accepts POST /api/inference-jobs/<id>/complete/ with `mask_image`, 400 on a second completion."""

    def __init__(self):
        self.done = {}
        outer = self

        class H(BaseHTTPRequestHandler):
            def log_message(self, *a):
                pass

            def do_POST(self):
                parts = self.path.strip("/").split("/")
                body = self.rfile.read(int(self.headers["Content-Length"]))
                ok = len(parts) == 4 and parts[:2] == ["api", "inference-jobs"] and parts[3] == "complete"
                _, files = parse_multipart(self.headers["Content-Type"], body)
                if not ok or "mask_image" not in files or parts[2] in outer.done:
                    self.send_response(400)
                    self.end_headers()
                    return
                outer.done[parts[2]] = files["mask_image"]
                self.send_response(200)
                self.send_header("Content-Length", "2")
                self.end_headers()
                self.wfile.write(b"{}")

        self.srv = ThreadingHTTPServer(("127.0.0.1", 0), H)
        self.url = f"http://127.0.0.1:{self.srv.server_address[1]}"
        threading.Thread(target=self.srv.serve_forever, daemon=True).start()


def _png(a):
    buf = io.BytesIO()
    Image.fromarray(a, "RGB").save(buf, format="PNG")
    return buf.getvalue()


def _post(url, fields, files, token=TOKEN):
    ctype, body = encode_multipart(files, fields)
    req = urllib.request.Request(url, data=body, method="POST", headers={"Content-Type": ctype, "X-ORCH-TOKEN": token})
    try:
        with urllib.request.urlopen(req, timeout=10) as r:
            return r.status, json.loads(r.read())
    except urllib.error.HTTPError as e:
        return e.code, json.loads(e.read() or b"{}")


def _fake_slot(calls, gate):
    def predict_batch(images):   # "network": class = the image's top-left red value mod 3, 8x8 masks
        calls.append(len(images))
        gate.wait(timeout=5.0)   # the first forward lasts until the test says every upload has been queued
        return [np.full((8, 8), int(a[0, 0, 0]) % 3, np.uint8) for a in images]
    return ModelSlot(predict_batch, 3)


@pytest.fixture
def service():
    be, calls, gate = FakeBackend(), [], threading.Event()
    gate.set()
    w = Worker({7: _fake_slot(calls, gate)}, be.url, TOKEN, max_batch=8, batch_wait_s=0.02)
    w.gate = gate
    srv = w.serve("127.0.0.1", 0)
    threading.Thread(target=srv.serve_forever, daemon=True).start()
    yield w, be, f"http://127.0.0.1:{srv.server_address[1]}", calls
    srv.shutdown()
    w.stop()
    be.srv.shutdown()


def test_multipart_round_trip():
    ctype, body = encode_multipart({"input_image": ("a.png", b"\x89PNG\r\n\x00\xff--x", "image/png")}, {"job_id": "j1", "n": "3"})
    fields, files = parse_multipart(ctype, body)
    assert fields == {"job_id": "j1", "n": "3"} and files["input_image"] == ("a.png", b"\x89PNG\r\n\x00\xff--x")


def test_contract_202_then_completion_with_mask_png(service):
    w, be, url, calls = service
    img = np.zeros((20, 30, 3), np.uint8)
    img[0, 0, 0] = 5                                       # -> class 2
    code, body = _post(url + "/enqueue/", {"job_id": "abc-1", "vision_model_id": "7"}, {"input_image": ("x.png", _png(img), "image/png")})
    assert code == 202 and body["job_id"] == "abc-1"       # the caller only checks for 202 (views.py:109)
    for _ in range(200):
        if "abc-1" in be.done:
            break
        time.sleep(0.02)
    name, png = be.done["abc-1"]
    got = np.array(Image.open(io.BytesIO(png)).convert("RGB"))
    assert name.endswith(".png") and got.shape == (8, 8, 3) and (got == default_palette(3)[2]).all()
    assert w.stats["completed"] == 1 and w.stats["failed"] == 0


def test_rejections(service):
    w, be, url, calls = service
    f = {"input_image": ("x.png", _png(np.zeros((4, 4, 3), np.uint8)), "image/png")}
    assert _post(url + "/enqueue/", {"job_id": "a", "vision_model_id": "7"}, f, token="wrong")[0] == 403
    assert _post(url + "/enqueue/", {"job_id": "a"}, f)[0] == 400                       # vision_model_id missing
    assert _post(url + "/enqueue/", {"job_id": "a", "vision_model_id": "7"}, {})[0] == 400   # no image
    assert _post(url + "/enqueue/", {"job_id": "a", "vision_model_id": "99"}, f)[0] == 404   # model not loaded
    assert _post(url + "/nowhere/", {"job_id": "a", "vision_model_id": "7"}, f)[0] == 404
    assert w.stats["received"] == 0 and not calls


def test_concurrent_jobs_share_a_forward(service):
    w, be, url, calls = service
    def send(i):
        img = np.zeros((6, 6, 3), np.uint8)
        img[0, 0, 0] = i
        assert _post(url + "/enqueue/", {"job_id": f"job{i}", "vision_model_id": "7"},
                     {"input_image": ("x.png", _png(img), "image/png")})[0] == 202
    w.gate.clear()                                         # hold the first forward open ...
    ts = [threading.Thread(target=send, args=(i,)) for i in range(12)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert w.stats["received"] == 12
    w.gate.set()                                           # ... until all 12 uploads sit in the queue
    for _ in range(300):
        if len(be.done) == 12:
            break
        time.sleep(0.02)
    assert len(be.done) == 12 and sum(calls) == 12
    assert max(calls) > 1 and max(calls) <= 8              # cross-job batching, bounded by max_batch
    for i in range(12):                                    # every job got ITS mask back
        got = np.array(Image.open(io.BytesIO(be.done[f"job{i}"][1])).convert("RGB"))
        assert (got == default_palette(3)[i % 3]).all()
    h = json.loads(urllib.request.urlopen(url + "/health", timeout=5).read())
    assert h["completed"] == 12 and h["largest_batch"] == max(calls)
