# Two rocprofv3 --pmc passes (SQ set; TCP/TA/TCC set) over a short bench run, summarised per kernel into
# gpurun_out/pmc_<tag>_summary.json (+ the raw per-kernel totals as text). Usage: tools/gpu_pmc_summary.sh <tag> [bench args]
set -e
cd $GRAFT_REPO_ROOT
TAG="$1"; shift
bash tools/gpu_pmc.sh ${TAG}_sq "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "$@" > gpurun_out/pmc_${TAG}_sq.txt
bash tools/gpu_pmc.sh ${TAG}_mem "TA_TA_BUSY TCC_HIT TCC_MISS TCP_GATE_EN1 TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ TCP_TOTAL_CACHE_ACCESSES GRBM_GUI_ACTIVE" "$@" > gpurun_out/pmc_${TAG}_mem.txt
python3 - "$TAG" <<'PY'
import sys, json, re, os
tag = sys.argv[1]
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
def parse(path):
    out, cur = {}, None
    for line in open(path):
        if not line.startswith(" "):
            cur = line.strip(); out[cur] = {}
        else:
            m = re.match(r"\s+(\S+)\s+total (\S+)\s+per-dispatch (\S+)\s+dispatches (\d+)", line)
            if m and cur: out[cur][m.group(1)] = (float(m.group(2)), int(m.group(4)))
    return out
sq, mem = parse(f"{root}/pmc_{tag}_sq.txt"), parse(f"{root}/pmc_{tag}_mem.txt")
ks = []
for k in sq:
    s, m = sq[k], mem.get(k, {})
    if "SQ_WAVE_CYCLES" not in s or not k.startswith(("pt::", "void pt::")): continue
    g = lambda d, c: d.get(c, (0.0, 0))[0]
    r = lambda a, b: round(a / b, 3) if b else None
    ks.append({"kernel": k, "dispatches": s["SQ_WAVE_CYCLES"][1],
               "valu_issue_share_of_wave_cycles": r(g(s, "SQ_ACTIVE_INST_VALU"), g(s, "SQ_WAVE_CYCLES")),
               "active_lane_fraction": r(g(s, "SQ_THREAD_CYCLES_VALU"), 64.0 * g(s, "SQ_ACTIVE_INST_VALU")),
               "wait_inst_any_share_of_wave_cycles": r(g(s, "SQ_WAIT_INST_ANY"), g(s, "SQ_WAVE_CYCLES")),
               "valu_insts_per_vmem_inst": r(g(s, "SQ_INSTS_VALU"), g(s, "SQ_INSTS_VMEM")),
               "l2_hit_rate": r(g(m, "TCC_HIT"), g(m, "TCC_HIT") + g(m, "TCC_MISS")),
               "tcp_pending_stall_share": r(g(m, "TCP_PENDING_STALL_CYCLES"), g(m, "TCP_GATE_EN1")),
               "ta_busy_share": r(g(m, "TA_TA_BUSY"), g(m, "TCP_GATE_EN1")),
               "tcp_miss_share_of_accesses": r(g(m, "TCP_TCC_READ_REQ"), g(m, "TCP_TOTAL_CACHE_ACCESSES")),
               "sq_wave_cycles_total": g(s, "SQ_WAVE_CYCLES")})
ks.sort(key=lambda d: -d["sq_wave_cycles_total"])
json.dump({"tag": tag, "command": "tools/gpu_pmc_summary.sh (two rocprofv3 --pmc passes over `bench.py --steps 1 --warmup 0`: SQ set, TCP/TA/TCC set)",
           "notes": "ratios of raw counter totals; raw per-kernel totals in the _sq.txt / _mem.txt files beside this one", "kernels": ks},
          open(f"{root}/pmc_{tag}_summary.json", "w"), indent=1)
for d in ks[:8]: print(d)
PY
