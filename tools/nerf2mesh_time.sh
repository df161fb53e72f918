#!/usr/bin/env bash
# f2 timing ON THE GPU BOX: train a short synthetic run to get reference-format files, then time the dense-grid query
# of `python -m hbr_amd.nerf2mesh` at 256^3 (the reference's resolution) and 512^3 (BASELINE config 5).
set -uo pipefail
TAG="${1:?tag}"; ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
W=/tmp/n2m_$$; mkdir -p $W; cd $W
PYTHONPATH="$ROOT" python3 -m hbr_amd.train_hash2 --synthetic 262144 --num_batch 16000 --num_samples 64 --num_epochs 4 --steps 60 --write --model_name n2m --out_dir $W/res > "$OUT/n2m_train.log" 2>&1
for res in 256 512; do
  for prec in fp32 bf16; do
    PYTHONPATH="$ROOT" python3 -m hbr_amd.nerf2mesh --bound_pth bounds_model.npy --ckpt_name n2m --resolution $res --precision $prec --out $W/grid_$res.npy 2>&1 | tail -1 | sed "s/^/res=$res precision=$prec  /" | tee -a "$OUT/nerf2mesh_time.txt"
    rm -f $W/grid_$res.npy
  done
done
rm -rf $W
