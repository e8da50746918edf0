# alternating runs of bench.py on one box under different kernel switches: LEGS="default NO_WSD_ST" bash tools/ab_env.sh
cd $GRAFT_REPO_ROOT
for i in 1 2; do
for L in ${LEGS:-default WSD32 WS32}; do
  unset CPNATIVE_WS32 CPNATIVE_WSD32 CPNATIVE_NO_WSD_ST CPNATIVE_NO_PROJ_FUSED CPNATIVE_MATERIALIZE_U8 CPNATIVE_TN16 CPNATIVE_NO_WSK CPNATIVE_TN_W4
  [ $L = WSD32 ] && export CPNATIVE_WSD32=1
  [ $L = WS32 ] && export CPNATIVE_WS32=1
  [ $L = NO_WSD_ST ] && export CPNATIVE_NO_WSD_ST=1
  [ $L = NO_PROJ_FUSED ] && export CPNATIVE_NO_PROJ_FUSED=1
  [ $L = MATERIALIZE_U8 ] && export CPNATIVE_MATERIALIZE_U8=1
  [ $L = TN16 ] && export CPNATIVE_TN16=1
  [ $L = NO_WSK ] && export CPNATIVE_NO_WSK=1
  [ $L = TN_W4 ] && export CPNATIVE_TN_W4=1
  python bench.py --no_cpu_baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
pk=d['roofline']['per_kernel']
sp=d.get('steps_spread', {})
print('%-10s %.3f ms/step (median %.3f, min %.3f)  ' % ('$L', d['ms_per_step'], sp.get('median_ms', 0), sp.get('min_ms', 0)) + '  '.join('%s %.1f' % (k, v['avg_us']) for k, v in pk.items()))"
done
done
