# a strong-scaled rank's share of cfg 3 on one GPU (rows per rank at N = 8, 4, 2); run on the GPU box
for M in 125000 250000 500000; do
  BENCH_M=$M python bench.py --steps 100 --no-cpu-baseline --survey-steps 0 --force-comm 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1])
c = d['config']
print('rows $M'.ljust(14), 'value %8.1f it/s  %6.3f ms per solve  %d it' % (d['value'], d['ms_per_step'], c['iterations_per_solve']), ' fdp %.3f lr %.3f gemm %.3f trial %.3f solve %.3f' % (d['roofline']['avg_launch_ms'], d['broyden_kernel']['avg_launch_ms'], d['residual_gemm']['avg_call_ms'], d['trial_residual']['avg_call_ms'], d['solve_kernel']['avg_launch_ms']))"
done
