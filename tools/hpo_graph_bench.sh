N=${1:-96}
cd $GRAFT_REPO_ROOT
for G in "" "--graph"; do for P in 1 2; do
  D=$(mktemp -d); S=$(date +%s.%N)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node=$P --master-addr 127.0.0.1 --master-port $((29800 + P)) \
      -m contrastiveprosthetics_amd.train --hpo_pack --crossval_size $N --crossval_epochs 1 --final_epochs 0 --batch_size 8 \
      --synthetic --no_adabn --no_checkpoint --data_dir $D --checkpoint_dir $D $G > $D/log.txt 2>&1 || { tail -5 $D/log.txt; exit 1; }
  E=$(date +%s.%N)
  python -c "
import numpy as np
v = np.load('$D/cross_val_values.npy'); print('graph=[$G] $P process(es): $N configurations in %.1f s, best val acc %.4f' % ($E - $S, np.nanmax(v[:, 1])))"
  rm -rf $D
done; done
