# a strong-scaled rank's share of cfg 3 on one GPU (rows per rank at N = 8, 4, 2); run on the GPU box
# usage: bash scripts/run_small.sh [extra bench.py args, e.g. --variant 4096]
for M in 125000 250000 500000; do
  BENCH_M=$M python bench.py --steps 100 --no-cpu-baseline --survey-steps 0 --force-comm "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1])
c = d['config']
k = {o['kernel'].split('<')[0].split('::')[-1].split(' ')[0]: o for o in (d.get('roofline'), d.get('broyden_kernel'), d.get('jtj_kernel')) if o}
ms = lambda name: next((o['avg_launch_ms'] for kk, o in k.items() if name in kk), float('nan'))
print('rows $M'.ljust(14), 'value %8.1f it/s  %6.3f ms per solve  %d it' % (d['value'], d['ms_per_step'], c['iterations_per_solve']), ' fdp %.3f lr %.3f gemm %.3f trial %.3f solve %.3f' % (ms('k_jtj_fdp'), ms('k_broyden_lr'), d['residual_gemm']['avg_call_ms'], d['trial_residual']['avg_call_ms'], d['solve_kernel']['avg_launch_ms']))"
done
