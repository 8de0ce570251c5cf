# the hand-over point of the re-judging step's first pass (evaluations) on the lln step, reference mode (GPU box)
for cap in ${@:-24 32 48 64 96 128 160}; do
  RT_REJ_FIRST_PASS_NFEV=$cap timeout -k 10 100 python bench.py --no-cpu-baseline --no-extra-configs --no-other-mode --steps 20 > gpurun_out/sweep.json || exit 1
  python -c "
import json;d=json.load(open('gpurun_out/sweep.json'));k=d['kernel_ms_per_step'];print('first pass up to $cap evaluations: step %.2f rejudge %.2f paths %d' % (d['ms_per_step'],k['rejudge'],d['config']['valid_paths_rank0']))"
done
