#!/bin/bash
# Wall time of the reference's random search (code/go.sh: --batch_size=8, 1 epoch per configuration) packed with
# 1, 2 and 4 processes on ONE MI355X (the GPU pool allows at most 6 processes on a card).  usage: tools/hpo_bench.sh [configs]
N=${1:-24}
cd "$(dirname "$0")/.."
for P in 1 2 4; do
  D=$(mktemp -d)
  S=$(date +%s.%N)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node=$P --master-addr 127.0.0.1 --master-port $((29700 + P)) \
      -m contrastiveprosthetics_amd.train --hpo_pack --crossval_size $N --crossval_epochs 1 --final_epochs 0 --batch_size 8 \
      --synthetic --no_adabn --no_checkpoint --data_dir $D --checkpoint_dir $D > $D/log.txt 2>&1 || { tail -5 $D/log.txt; exit 1; }
  E=$(date +%s.%N)
  python - <<PY
import numpy as np
v = np.load("$D/cross_val_values.npy")
print(f"{$P} process(es): {$N} configurations in {$E - $S:6.1f} s  (incl. start-up)  best val acc {np.nanmax(v[:, 1]):.4f}")
PY
  rm -rf $D
done
