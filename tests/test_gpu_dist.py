"""GPU (-m gpu): `bench.py --gpus 2` rehearsed on ONE card -- both ranks on device 0 (VITSEG_LOCAL_DEVICE), gloo in place of
RCCL (VITSEG_DIST_BACKEND) -- so that the N > 1 control flow the driver's scaling run takes (launcher, batch split, the
data-parallel training side path with its bucketed gradient all-reduce and the no_sync() twin) has run end to end on
hardware at least once.  No scaling figure is read from it."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_on_one_gpu_exercises_split_and_gradient_exchange():
    env = dict(os.environ, VITSEG_LOCAL_DEVICE="0", VITSEG_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", "4", "--dist-train-batch", "2", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout            # rank 0's line, once
    out = lines[0]
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 8 and out["scaling"] == "weak"
    sp = out["inference_split"]
    assert sp["ranks_seen"] == 2 and sp["mask_pixels_per_rank"] == [4 * 512 * 512] * 2
    assert len(set(sp["mask_checksum_per_rank"])) == 2          # each rank ran ITS shard of the image stream, not the same one
    tr = out["train_bf16_path"]
    assert tr["ranks_seen"] == 2 and tr["backend"] == "gloo" and tr["global_batch"] == 4
    assert tr["ms_per_step"] > 0 and tr["ms_per_step_no_sync"] > 0 and tr["gradient_bytes_per_step"] > 300e6
    assert tr["all_reduce_messages_per_step"] >= 2 and tr["final_loss"] == tr["final_loss"]   # bucketed; loss is not NaN
