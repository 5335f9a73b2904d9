cd $GRAFT_REPO_ROOT
one() { python bench.py --no-cpu --no-iso "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('   ', d['config']['name'], ' '.join('%s %.3f'%(k['kernel'].split('<')[0].replace('bfd::','').replace('bf_render_kernel','tail'),k['ms_per_step']) for k in r['kernels']), '| step', d['ms_per_step'], 'serial', d['ms_per_step_serial'])"; }
for cfg in c2 c3 c4shard c5; do
  echo "== $cfg default"; one --config $cfg
  for it in 1 2 3 4; do echo "BF_ROLL_ITERS=$it"; BF_ROLL_ITERS=$it one --config $cfg; done
  for lv in 524288 3145728; do echo "BF_ROLL_LIVE=$lv"; BF_ROLL_LIVE=$lv one --config $cfg; done
done
