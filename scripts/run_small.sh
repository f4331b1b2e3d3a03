# a strong-scaled rank's share of cfg 3 on one GPU (rows per rank at N = 8, 4, 2), with and without a one-rank RCCL
# communicator (--force-comm); run on the GPU box.  usage: bash scripts/run_small.sh [extra bench.py args]
# Per line: rate, ms per solve (mean and min / median / max over the timed steps), passes, rounds by kind, library launches per
# round, all-reduce calls, the event-timed per-kernel averages and the per-solve time split.
for M in ${SMALL_ROWS:-125000 250000 500000}; do
 for COMM in "--force-comm" ""; do
  BENCH_M=$M python bench.py --steps ${SMALL_STEPS:-200} --no-cpu-baseline --no-host-callback --survey-steps 0 $COMM "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1])
c = d['config']
k = {o['kernel'].split('<')[0].split('::')[-1].split(' ')[0]: o for o in (d.get('roofline'), d.get('broyden_kernel'), d.get('jtj_kernel')) if o}
ms = lambda name: next((o['avg_launch_ms'] for kk, o in k.items() if name in kk), float('nan'))
mm = d.get('ms_per_step_uninstrumented_min_median_max') or d['ms_per_step_min_median_max']
ar = c['allreduce_per_solve']; ts = c['time_split_ms_per_solve']; r = c['rounds_per_solve']; l = c['library_launches_per_round']
print(('rows $M ' + ('rccl x1' if '$COMM' else 'no comm')).ljust(22), '%8.1f it/s  %6.3f ms/solve (uninstrumented min %.3f med %.3f max %.3f)  %d it %g passes' % (d['value'], d['ms_per_step'], mm[0], mm[1], mm[2], c['iterations_per_solve'], c['passes_per_solve']))
print(' ' * 22, 'rounds refresh %g broyden %g resolve %g  launches/round %s  allreduce packed %g sweep %g scalar %g  stall %s' % (r['refresh'], r['broyden'], r['resolve'], [None if v is None else round(v, 1) for v in l.values()], ar['packed_calls'], ar['sweep_calls'], ar['scalar_calls'], (c.get('rccl_stall_probe') or {}).get('observed')))
print(' ' * 22, 'kernels: fdp %.3f lr %.3f gemm %.3f trial %.3f solve %.3f   per solve: caller %.3f library %.3f (fd %.3f sweep %.3f solve %.3f) wall %.3f' % (ms('k_jtj_fdp'), ms('k_broyden_lr'), d['residual_gemm']['avg_call_ms'], d['trial_residual']['avg_call_ms'], d['solve_kernel']['avg_launch_ms'], ts['caller_kernels'], ts['library_kernels'], ts['jtj_fd_kernel'], ts['broyden_sweep'], ts['solve_kernel'], ts['total_wall']))"
 done
done
