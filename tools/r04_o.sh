cd $GRAFT_REPO_ROOT
one() { python bench.py --no-cpu --no-iso "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('   ', d['config']['name'], ' '.join('%s %.3f'%(k['kernel'].split('<')[0].replace('bfd::','').replace('bf_render_kernel','tail'),k['ms_per_step']) for k in r['kernels']), '| step', d['ms_per_step'], 'serial', d['ms_per_step_serial'])"; }
for cfg in c2 c5 c3 c4shard; do
  for v in 44 32 24 16 8; do echo "BF_TRACE_REFILL=$v"; BF_TRACE_REFILL=$v one --config $cfg; done
  echo "BF_TRACE_REFILL=24 BF_TRACE_STRAGGLERS=8"; BF_TRACE_REFILL=24 BF_TRACE_STRAGGLERS=8 one --config $cfg
  echo "BF_TRACE_REFILL=24 BF_TRACE_STRAGGLERS=16"; BF_TRACE_REFILL=24 BF_TRACE_STRAGGLERS=16 one --config $cfg
done
