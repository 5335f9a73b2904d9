cd $GRAFT_REPO_ROOT
one() { python bench.py --no-cpu --no-iso "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(' '.join('%s %.3f'%(k['kernel'].split('<')[0].replace('bfd::','').replace('bf_render_kernel','tail'),k['ms_per_step']) for k in r['kernels']), '| step', d['ms_per_step'], 'serial', d['ms_per_step_serial'])"; }
echo "base"; one; one
echo "BF_QUANT_BVH=1"; BF_QUANT_BVH=1 one; BF_QUANT_BVH=1 one
echo "BF_QUANT_BVH=1 BF_TRACE_WAVES=6"; BF_QUANT_BVH=1 BF_TRACE_WAVES=6 one
echo "BF_TRACE_WAVES=4"; BF_TRACE_WAVES=4 one
echo "BF_TRACE_REFILL=32"; BF_TRACE_REFILL=32 one
echo "BF_TRACE_REFILL=52"; BF_TRACE_REFILL=52 one
echo "BF_TRACE_STRAGGLERS=6"; BF_TRACE_STRAGGLERS=6 one
echo "BF_TRACE_STRAGGLERS=20"; BF_TRACE_STRAGGLERS=20 one
echo "c5 base"; one --config c5
echo "c5 BF_QUANT_BVH=1"; BF_QUANT_BVH=1 one --config c5
