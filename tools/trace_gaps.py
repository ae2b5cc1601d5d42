"""Where a step's wall time goes on the GPU: kernel trace of a short bench run, one line per kernel of the last step (start offset,
duration), so that idle stretches between launches show.  GPU box only: python tools/trace_gaps.py [workload]"""
import csv
import glob
import os
import subprocess
import sys

wl = sys.argv[1] if len(sys.argv) > 1 else "pacbio_d150_msa150"
out = "/tmp/trace_gaps"
subprocess.run(f"rm -rf {out}; cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace -d {out} -o t --output-format csv -- python {os.getcwd()}/bench.py --steps 3 --warmup 1 --cpu-sample 0 --pcie-steps 0 --workload {wl} > /tmp/trace_gaps.log 2>&1",
               shell=True, check=False)
f = glob.glob(out + "/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]) for r in rows))
# the last step = from the last cw_setup_need_kernel on
idx = [i for i, k in enumerate(ks) if "cw_setup_need" in k[2]]
i0 = idx[-2]
t0 = ks[i0][0]
prev_end = None
for s, e, n in ks[i0 : idx[-1] + 3]:
    print(f"{(s - t0) / 1e6:9.3f} ms  +{(e - s) / 1e6:8.3f} ms  {n}")
