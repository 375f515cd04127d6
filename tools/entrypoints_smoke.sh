# the reference's script entry points on the GPU box (random weights, synthetic data)
set -e
python3 - <<'PY'
import numpy as np
from PIL import Image
Image.fromarray(np.random.RandomState(0).randint(0, 256, (300, 400, 3), dtype=np.uint8), "RGB").save("/tmp/in.png")
PY
cd model/CE
python3 testViTModel.py /tmp/in.png --model-id 1 --num-classes 17 --out /tmp/mask.png
python3 testViTModel.py /tmp/in.png --model-id 1 --num-classes 17 --precision fp32x3
python3 trainCurrentViTmodel.py --model-id 1 --epochs 2 --batches 3 --batch-size 4 2>&1 | tail -4
