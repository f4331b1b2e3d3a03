#!/bin/bash
# A/B of the per-pass launch sequence on the GPU box: round 2's sequence, the default (finish in the
# solve prologue, unpack in the slab reduction), and the opt-in "last workgroup finishes" tails. cfg 3, cfg 2 and a
# strong-scaled rank's 125 000 rows. usage: bash scripts/ab_tails.sh > gpurun_out/ab_tails.txt
for cfg in "--config cfg3 --steps 40 --warmup 5 --survey-steps 0" "--config cfg2" "--config cfg3 --rows 125000 --steps 100 --survey-steps 0"; do
  for v in 4096 0 16384 32768 49152; do
    python bench.py $cfg --no-cpu-baseline --no-host-callback --variant $v 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1])
names = {4096: 'round-2 sequence', 0: 'default (merged into consumers)', 16384: '+ sweep tail', 32768: '+ sumsq/decide tail', 49152: '+ both tails'}
c = d['config']
print('$cfg'.split(' --steps')[0].ljust(28), names[$v].ljust(34), 'value %9.1f it/s   %7.3f ms per solve   %s' % (d['value'], d['ms_per_step'], ('%.1f us per round' % c['us_per_round']) if 'us_per_round' in c else ''))"
  done
done
