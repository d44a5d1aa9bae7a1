#!/bin/bash
# Where the attention kernels' wave-cycles go (non-perturbing: SQ counters, rocprofv3 --pmc in their own passes; MI355X guide, PMC slots):
# SQ_WAIT_ANY = parked on s_waitcnt / barrier, SQ_WAIT_INST_ANY = issue stall (pipe busy / dependency), SQ_ACTIVE_INST_ANY = issuing.
#   on the GPU: tools/attn_wave_cycles.sh     -> gpurun_out/attn_wave_cycles.txt (record: profiles/r06_attn_fwd_steptrace.txt, part 4)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/awc; rm -rf $O; mkdir -p $O
B="python tools/attn_bench.py 20"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES --output-format csv -d $O/p1 -o run -- $B > $O/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $O/p2 -o run -- $B > $O/p2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS GRBM_GUI_ACTIVE --output-format csv -d $O/p3 -o run -- $B > $O/p3.log 2>&1
python - <<'PY' > gpurun_out/attn_wave_cycles.txt
import csv, glob, collections, re
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
for p in ("p1", "p2", "p3"):
    for f in glob.glob(f"gpurun_out/awc/{p}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(attn_\w+)", r["Kernel_Name"])
            if not m: continue
            k = m.group(1)
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
for k, c in tot.items():
    n = max(cnt[k], 1)
    print(f"== {k}: {n} launches; per launch:")
    wc = c.get("SQ_WAVE_CYCLES", 0) / n
    for name in sorted(c):
        v = c[name] / n
        print(f"   {name:28s} {v:14.0f}" + (f"   {100 * v / wc:5.1f} % of wave-cycles" if wc and name.startswith(("SQ_WAIT", "SQ_ACTIVE", "SQ_INST_CYCLES")) else ""))
PY
tail -5 $O/p1.log | head -3
cat gpurun_out/attn_wave_cycles.txt
