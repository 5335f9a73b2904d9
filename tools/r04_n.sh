cd $GRAFT_REPO_ROOT
one() { python bench.py --no-cpu --no-iso "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('   ', d['config']['name'], ' '.join('%s %.3f'%(k['kernel'].split('<')[0].replace('bfd::','').replace('bf_render_kernel','tail'),k['ms_per_step']) for k in r['kernels']), '| step', d['ms_per_step'], 'serial', d['ms_per_step_serial'])"; }
for cfg in c2 c5; do
  echo "== $cfg default"; one --config $cfg; one --config $cfg
  for v in 4 12; do echo "BF_SHADE_CHAIN=$v"; BF_SHADE_CHAIN=$v one --config $cfg; done
  for v in 0 24; do echo "BF_CHAIN_MIN=$v"; BF_CHAIN_MIN=$v one --config $cfg; done
  for v in 32 52; do echo "BF_TRACE_REFILL=$v"; BF_TRACE_REFILL=$v one --config $cfg; done
  echo "--streams 3"; one --config $cfg --streams 3
  echo "--streams 1"; one --config $cfg --streams 1
done
