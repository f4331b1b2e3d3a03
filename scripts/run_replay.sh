# ONE rank of an R-rank strong-scaled cfg 3 on this GPU, on the GLOBAL problem's trajectory (bench.py --replay-ranks: the
# R-shard solve is run once in-process and rank 0's all-reduce totals are recorded; the timed solves are rank 0's shard alone
# with every exchange replaced by the recorded total). Second line per R: every exchange also through a one-rank ncclAllReduce.
# REPLAY_LATENCIES="10 20 40": also with every exchange charged that many microseconds (a MODEL of an N-rank all-reduce's latency).
# usage: bash scripts/run_replay.sh [extra bench.py args]
for R in ${REPLAY_RANKS:-8 4 2}; do
 for COMM in "" "--force-comm" $(for L in ${REPLAY_LATENCIES:-}; do echo "--replay-latency-us=$L"; done); do
  python bench.py --steps ${REPLAY_STEPS:-200} --no-cpu-baseline --no-host-callback --replay-ranks $R $COMM "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1])
c = d['config']
k = {o['kernel'].split('<')[0].split('::')[-1].split(' ')[0]: o for o in (d.get('roofline'), d.get('broyden_kernel'), d.get('jtj_kernel')) if o}
ms = lambda name: next((o['avg_launch_ms'] for kk, o in k.items() if name in kk), float('nan'))
mm = d.get('ms_per_step_uninstrumented_min_median_max') or d['ms_per_step_min_median_max']
ar = c['allreduce_per_solve']; ts = c['time_split_ms_per_solve']; r = c['rounds_per_solve']; l = c['library_launches_per_round']
print(('rank 0 of $R ' + ('+rccl x1' if '$COMM' == '--force-comm' else ('+' + '$COMM'.split('=')[-1] + ' us/xchg' if '$COMM' else 'replay  '))).ljust(22), '%8.1f it/s  %6.3f ms/solve (min %.3f med %.3f max %.3f)  %d rows  %d it %g passes  grouped %d-shard solve %.1f ms' % (d['value'], d['ms_per_step'], mm[0], mm[1], mm[2], c['m_per_gpu'], c['iterations_per_solve'], c['passes_per_solve'], $R, c['replay']['grouped_solve_wall_ms']))
print(' ' * 22, 'rounds refresh %g broyden %g resolve %g  launches/round %s  allreduce packed %g sweep %g scalar %g' % (r['refresh'], r['broyden'], r['resolve'], [None if v is None else round(v, 1) for v in l.values()], ar['packed_calls'], ar['sweep_calls'], ar['scalar_calls']))
print(' ' * 22, 'kernels: fdp %.3f lr %.3f gemm %.3f trial %.3f solve %.3f   per solve: caller %.3f library %.3f (fd %.3f sweep %.3f solve %.3f) wall %.3f' % (ms('k_jtj_fdp'), ms('k_broyden_lr'), d['residual_gemm']['avg_call_ms'], d['trial_residual']['avg_call_ms'], d['solve_kernel']['avg_launch_ms'], ts['caller_kernels'], ts['library_kernels'], ts['jtj_fd_kernel'], ts['broyden_sweep'], ts['solve_kernel'], ts['total_wall']))"
 done
done
