# L2 request counters of the denoiser's weight-gradient launches (three eager text steps at config 5's size), ring kernel vs the
# 128 x 128-tile kernel on one box: what the CUs request from L2 is what they stage into LDS.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/tnl2; rm -rf $O; mkdir -p $O
for x in 0 1; do
  export TDM_TN_RING=$x
  timeout -k 10 300 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/p$x -- python tools/text_steps.py 0.1 3 > $O/log$x.txt 2>&1
done
unset TDM_TN_RING
python - "$O" <<'PY'
import csv, glob, sys, re
from collections import defaultdict
root = sys.argv[1]
for x in (0, 1):
    agg = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
    for f in glob.glob(f"{root}/p{x}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "gemm_tn" not in n: continue
            k = re.sub(r"\(.*$", "", n.replace("void ", ""))[:40]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "TCC_REQ_sum": cnt[k] += 1
    for k, d in agg.items():
        print(f"TDM_TN_RING={x} {k:40s} launches {cnt[k]:3d}  TCC_REQ {d['TCC_REQ_sum']:.3e}  HIT {d['TCC_HIT_sum']:.3e}  MISS {d['TCC_MISS_sum']:.3e}  per step (3 steps): REQ {d['TCC_REQ_sum'] / 3:.3e}")
PY
rm -rf $O/p0 $O/p1
