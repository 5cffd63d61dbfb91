"""Mean per dispatch of every counter, per kernel, from the rocprofv3 --pmc CSVs under gpurun_out/pmc_<tag>/.
Dispatches that did (almost) nothing -- the gated instance of k_fuse that returns at once -- are left out of the mean."""
import csv, glob, sys, collections, re
tag = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(f'gpurun_out/pmc_{tag}/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        m = re.search(r'(k_\w+|rocprim|rocclr\w*)', k)
        acc[(m.group(1) if m else k[:40], r['Counter_Name'])].append(float(r['Counter_Value']))
want = sys.argv[2] if len(sys.argv) > 2 else None
for (k, c), v in sorted(acc.items()):
    if want is None or want == k:
        top = max(v)
        real = [x for x in v if x > 0.01 * top] or v
        print(f'{k},{c},{sum(real) / len(real)},{len(real)}')
