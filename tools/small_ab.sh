for r in 1 2 3; do for L in "" build/libcp_prev.so; do
CPNATIVE_LIB=${L:+$PWD/$L} python bench.py --no_cpu_baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); sb=d['small_batch']; print('lib=[${L:-new}]', 'b8 %.4f b32 %.4f' % (sb['b8']['ms_per_step'], sb['b32']['ms_per_step']), 'eval bf16 %.3f' % d['eval']['bf16']['ms'])"
done; done
