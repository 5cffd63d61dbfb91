"""Print step / fused-call / sort milliseconds of bench logs: python scripts/show_ms.py b1 b2 ... (gpurun_out/<name>.log)."""
import json, sys
for f in sys.argv[1:]:
    d = json.loads(open(f'gpurun_out/{f}.log').read().strip().split('\n')[-1])
    print(f, 'step', d['ms_per_step'], 'call', d['roofline']['kernel_ms'], 'sort', d['roofline']['sort_ms'])
