# the reference's script entry points on the GPU box (random weights, synthetic data); run from the repo root
set -e
ROOT=$(pwd)
python3 - <<'PY'
import numpy as np
from PIL import Image
Image.fromarray(np.random.RandomState(0).randint(0, 256, (300, 400, 3), dtype=np.uint8), "RGB").save("/tmp/in.png")
PY
W=$(mktemp -d)
cd $W && ln -s $ROOT/model model
cd $ROOT/model/CE
python3 testViTModel.py /tmp/in.png --model-id 1 --num-classes 17 --out /tmp/mask.png
python3 testViTModel.py /tmp/in.png --model-id 1 --num-classes 17 --precision fp32x3
(cd $W && python3 $ROOT/model/CE/trainCurrentViTmodel.py --model-id 1 --epochs 2 --batches 3 --batch-size 4 --ckpt-dir $W/logs/vit-model/version_1/checkpoints 2>&1 | tail -2)
(cd $W && python3 $ROOT/model/CE/createViTmodel.py --hidden-size 512 --layers 8 --heads 8 --epochs 2 --batches 3 --version 2 2>&1 | tail -2)
(cd $W && python3 $ROOT/model/CE/datasetTestViTmodel.py --ids 1 --num-classes 2 --num-batches 2 2>&1 | tail -2 && head -3 test/ID1P16H512A8/ID1P16H512A8_metrics.csv)
(cd $W && python3 $ROOT/model/PAED/ViTscript.py --patch-size 16 --hidden-size 512 --layers 8 --heads 8 --epochs 2 --batches 3 --version 3 2>&1 | tail -2)
(cd $W && python3 $ROOT/model/PAED/ViTscriptUp.py --patch-size 16 --hidden-size 512 --layers 8 --heads 8 --epochs 1 --batches 2 --version 4 2>&1 | tail -2)
(cd $W && python3 $ROOT/model/PAED/ViTscriptUp.py --patch-size 16 --hidden-size 512 --layers 8 --heads 8 --epochs 2 --batches 2 --version 4 2>&1 | tail -2)
(cd $W && python3 $ROOT/model/PAED/ViTscriptTest.py --ids 3 --num-batches 2 2>&1 | tail -2)
echo "entry points ok"
