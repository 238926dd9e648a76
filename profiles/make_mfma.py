"""Matrix-pipe counters of the C3 forward per kernel from one rocprofv3 --pmc pass
(SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE), written to profiles/pmc_mfma.json
for bench.py's roofline.matrix_pipe.busy_per_kernel and printed as a table.
matrix_pipe_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs); the wait /
active columns are fractions of SQ_WAVE_CYCLES.
usage: python profiles/make_mfma.py <pmc_dir> [output.json]"""
import collections
import csv
import glob
import json
import os
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mvf::" not in r["Kernel_Name"]:
            continue
        name = r["Kernel_Name"].split("mvf::")[1].split("(")[0]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
print(f"{'kernel':42s} {'launches':>8s} {'gui_cycles/8':>12s} {'pipe_busy':>9s} {'mops_f32':>10s} {'mops_bf16':>10s} "
      f"{'wait_any':>8s} {'wait_inst':>9s} {'active':>7s}")
for k, cs in sorted(acc.items()):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    gui = m.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    wave = max(m.get("SQ_WAVE_CYCLES", 0.0), 1.0)
    row = dict(launches=len(next(iter(cs.values()))), gui_cycles=gui,
               matrix_pipe_busy=m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(gui * 1024.0, 1.0),
               mops_f32=m.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0), mops_bf16=m.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0),
               wait_any=m.get("SQ_WAIT_ANY", 0.0) / wave, wait_inst_any=m.get("SQ_WAIT_INST_ANY", 0.0) / wave,
               active_inst_any=m.get("SQ_ACTIVE_INST_ANY", 0.0) / wave)
    out[k] = row
    print(f"{k[:42]:42s} {row['launches']:8d} {gui:12.4g} {row['matrix_pipe_busy']:9.3f} {row['mops_f32']:10.4g} "
          f"{row['mops_bf16']:10.4g} {row['wait_any']:8.3f} {row['wait_inst_any']:9.3f} {row['active_inst_any']:7.3f}")
path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "pmc_mfma.json")
json.dump(out, open(path, "w"), indent=1, sort_keys=True)
