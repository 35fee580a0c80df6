#!/usr/bin/env python3
"""Integer-issue model of the proof_verify kernels: the `valu_issue` ceiling of bench.py, reproducible by hand.

    tools/valu_model.py profiles/r02_c          # reads  <prefix>_pmc.csv  <prefix>_ubench_valu_int.csv  <prefix>_isa_histogram.csv
                                                # writes <prefix>_counters.json  (what bench.py loads)  and prints the table

Per kernel:   issue cycles per launch = SQ_INSTS_VALU x cycles_per_inst,
              cycles_per_inst        = sum over opcode classes of  (share of the class in the kernel's hot-loop ISA)
                                       x (micro-benchmarked issue cost of the class at the kernel's waves per SIMD).
Issue costs come from tools/ubench/valu_int.hip (ns per wave-instruction per SIMD at 1 / 2 / 4 / 8 waves per SIMD, turned
into cycles with the clock the same micro-benchmark measured at one wave per SIMD).  The opcode shares come from
tools/isa_histogram.py: loop bodies of the kernel symbol + the whole text of the device functions it calls.
bench.py then reports  frac = sum_k issue_cycles_k / (1024 SIMDs x clock under load x seconds per step).
"""
import collections
import csv
import json
import sys

STAGE_OF = {"PairDist<BlsCurve>": "pairing_6lane", "PvMsmPart<BlsCurve>": "pv_msm_parts", "PvChallenge<BlsCurve>": "pv_challenge",
            "PvScalars<BlsCurve>": "pv_scalars", "PvFinish": "pv_finish", "PairMillerHalf<BlsCurve>": "pair_miller",
            "PairFinalDist<BlsCurve>": "pair_final_exp", "PairMillerBoth<BlsCurve>": "pair_miller_both",
            # round 5: the multi-scalar multiplication as kernels of their own (stages.hpp)
            "PvT1Chain<BlsCurve>": "pv_t1_chain", "PvVarMul<BlsCurve>": "pv_var_mul", "PvFixedChunk<BlsCurve>": "pv_fixed_chunks",
            "PvChains<BlsCurve>": "pv_chains"}
# the other three operations (tools/run_profile_ops.sh -> <prefix>_pmc_ops.csv): kernel -> stage name, per operation; SURVEY 8(d)'s
# algorithmic bytes per item (BLS12-381, L = 32, R = 8)
OPS = {"sign": {"alg_bytes": 1104, "kernels": {"VfIngest<BlsCurve>": "sg_ingest", "SgScalars<BlsCurve>": "sg_scalars", "SgMsmPart<BlsCurve>": "sg_msm_parts",
                                               "SgCombine<BlsCurve>": "sg_combine", "SgEmit<BlsCurve>": "sg_emit"}},
       "verify": {"alg_bytes": 1105, "kernels": {"VfIngest<BlsCurve>": "vf_ingest", "VfScalars<BlsCurve>": "vf_scalars", "VfVarMul<BlsCurve>": "vf_var_mul",
                                                 "VfFixedChunk<BlsCurve>": "vf_fixed_chunks", "VfCombine<BlsCurve>": "vf_combine", "PairDist<BlsCurve>": "pairing_6lane"}},
       "proof_gen": {"alg_bytes": 3136, "kernels": {"PgIngest<BlsCurve>": "pg_ingest", "PgScalars<BlsCurve>": "pg_scalars", "PgBPart<BlsCurve>": "pg_b_parts",
                                                    "PgBCombine<BlsCurve>": "pg_b_combine", "PgTables<BlsCurve>": "pg_tables", "PgVarPart<BlsCurve>": "pg_var_parts",
                                                    "PgFinalize<BlsCurve>": "pg_finalize", "PgEmit<BlsCurve>": "pg_emit"}}}


def load_occupancy(path):
    """waves per SIMD of every kernel from tools/kernel_meta.sh output (<prefix>_kernel_meta.txt): 512 registers per lane and
    SIMD, a wavefront takes vgpr + agpr of them (allocation granule 8); 1 when the file is missing"""
    import os
    import re
    occ = {}
    if not os.path.exists(path):
        return occ
    for line in open(path):
        m = re.match(r"\d*(\w+?)INS1_\d+(\w+?)E.*vgpr=(\d+) agpr=(\d+)", line.strip())
        if m:
            regs = -(-max(int(m.group(3)) + int(m.group(4)), 1) // 8) * 8
            occ["%s<%s>" % (m.group(1), m.group(2))] = max(1, min(8, 512 // regs))
    return occ
SIMPLE = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_lshlrev_b32", "v_ashrrev_i32",
          "v_mov_b32", "v_accvgpr_read_b32", "v_accvgpr_write_b32", "v_cndmask_b32", "v_not_b32", "v_max_i32", "v_min_i32", "v_max_u32",
          "v_min_u32", "v_readlane_b32", "v_writelane_b32", "v_readfirstlane_b32", "v_mul_i32_i24", "v_mul_u32_u24", "v_bfrev_b32"}


def load_ubench(path):
    t = collections.defaultdict(dict)
    for r in csv.DictReader(l for l in open(path) if not l.startswith("#")):
        t[r["op"]][(int(r["chains"]), int(r["waves_per_simd"]))] = (float(r["ns_per_wave_instr_per_simd"]), float(r["clock_ghz"]))
    return t


def class_costs(ub, waves):
    """cycles per wave-instruction per SIMD for the three classes at `waves` waves per SIMD (8 independent chains)"""
    w = min((1, 2, 4, 8), key=lambda x: abs(x - waves))
    ghz = lambda op: ub[op][(8, 1)][1]                     # clock of that op's one-wave run
    ns = lambda op: ub[op][(8, w)][0]
    simple_ops = ["v_add_u32", "v_sub_u32", "v_and_b32", "v_or_b32", "v_lshrrev_b32", "v_mov_b32"]
    vop3_ops = ["v_mul_lo_u32", "v_add3_u32", "v_lshl_add_u32", "v_lshl_or_b32", "v_and_or_b32", "v_bfe_u32", "v_alignbit_b32", "v_lshl_add_u64"]
    simple = sum(ns(o) * ghz(o) for o in simple_ops) / len(simple_ops)
    vop3 = sum(ns(o) * ghz(o) for o in vop3_ops) / len(vop3_ops)
    mix = "mix:3xv_mad_u64_u32+1xv_and_b32"
    mad = (4 * ns(mix) * ghz(mix) - ns("v_and_b32") * ghz("v_and_b32")) / 3.0    # the mad's share of the 3 + 1 mix
    carry = ns("v_add_co_u32+v_addc_co_u32") * ghz("v_add_co_u32+v_addc_co_u32")     # per instruction of the pair
    return {"mad": mad, "vop3": vop3, "simple": simple, "carry": carry}


def class_costs_ns(ub, waves):
    """the same four classes in NANOSECONDS per wave-instruction per SIMD (8 independent chains).  Time, not cycles: the
    chip clocks down as more wavefronts issue (the micro-benchmark's own clock column: 2.4 GHz at one wavefront per SIMD,
    1.3 - 1.6 at four), so a cost in cycles at one occupancy times a clock measured at another over-states what
    co-residency buys; a cost in nanoseconds is what the SIMD actually delivered."""
    w = min((1, 2, 4, 8), key=lambda x: abs(x - waves))
    ns = lambda op: ub[op][(8, w)][0]
    simple_ops = ["v_add_u32", "v_sub_u32", "v_and_b32", "v_or_b32", "v_lshrrev_b32", "v_mov_b32"]
    vop3_ops = ["v_mul_lo_u32", "v_add3_u32", "v_lshl_add_u32", "v_lshl_or_b32", "v_and_or_b32", "v_bfe_u32", "v_alignbit_b32", "v_lshl_add_u64"]
    mix = "mix:3xv_mad_u64_u32+1xv_and_b32"
    return {"mad": (4 * ns(mix) - ns("v_and_b32")) / 3.0, "vop3": sum(ns(o) for o in vop3_ops) / len(vop3_ops),
            "simple": sum(ns(o) for o in simple_ops) / len(simple_ops), "carry": ns("v_add_co_u32+v_addc_co_u32")}


def classify(op):
    base = op.replace("_e32", "").replace("_e64", "").replace("_sdwa", "").replace("_dpp", "")
    if base == "v_mad_u64_u32" or base == "v_mad_i64_i32":
        return "mad"
    if base in ("v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32", "v_subrev_co_u32", "v_subbrev_co_u32"):
        return "carry"
    if base in SIMPLE and not op.endswith("_e64"):
        return "simple"
    if base.startswith("v_cmp") and not op.endswith("_e64"):
        return "simple"
    return "vop3"


def model_ops(pre, ub, hist, occupancy, classify):
    """sign / verify / proof_gen from <prefix>_pmc_ops.csv: per kernel the counters of one 4096-item launch and the modelled
    issue cost of its opcode mix (ns per wave-instruction per SIMD at the occupancy its registers allow)"""
    import os
    path = pre + "_pmc_ops.csv"
    if not os.path.exists(path):
        return None
    pmc = collections.defaultdict(lambda: collections.defaultdict(dict))
    for r in csv.DictReader(l for l in open(path) if not l.startswith("#")):
        pmc[r["op"]][r["kernel"]][r["counter"]] = float(r["mean_per_launch"])
    out = {}
    any_hist = hist.get("PvT1Chain<BlsCurve>") or next(iter(hist.values()))
    for op, spec in OPS.items():
        ks = {}
        for k, stage in spec["kernels"].items():
            c = pmc.get(op, {}).get(k)
            if not c or "SQ_INSTS_VALU" not in c:
                continue
            h = hist.get(k) or any_hist
            tot = sum(h.values())
            share = collections.Counter()
            for o, nn in h.items():
                share[classify(o)] += nn / tot
            w = min(2, occupancy.get(k, 1))
            ns = sum(share[cl] * class_costs_ns(ub, w)[cl] for cl in share)
            ks[stage] = {"kernel": k, "valu_insts": c["SQ_INSTS_VALU"], "FETCH_SIZE_KiB": c.get("FETCH_SIZE", 0.0), "WRITE_SIZE_KiB": c.get("WRITE_SIZE", 0.0),
                         "waves": c.get("SQ_WAVES"), "waves_per_simd": w, "ns_per_inst_at_that_occupancy": round(ns, 4),
                         "ns_per_inst_if_waves_per_simd": {str(x): round(sum(share[cl] * class_costs_ns(ub, x)[cl] for cl in share), 4) for x in (1, 2, 4, 8)},
                         "own_isa_histogram": bool(hist.get(k)), "duration_ns_exclusive": c.get("duration_ns")}
        if ks:
            out[op] = {"alg_bytes_per_item": spec["alg_bytes"], "items_per_launch": 4096, "kernels": ks}
            print("%-10s %d kernels, %.4g VALU wave-instructions per 4096-item job, traffic %.1f MB" % (
                op, len(ks), sum(v["valu_insts"] for v in ks.values()),
                sum(2 * v["FETCH_SIZE_KiB"] + v["WRITE_SIZE_KiB"] for v in ks.values()) * 1024 / 1e6))
    return out or None


def main():
    pre = sys.argv[1]
    ub = load_ubench(pre + "_ubench_valu_int.csv")
    pmc = collections.defaultdict(dict)
    for r in csv.DictReader(l for l in open(pre + "_pmc.csv") if not l.startswith("#")):
        pmc[r["kernel"]][r["counter"]] = float(r["mean_per_launch"])
    hist = collections.defaultdict(collections.Counter)
    for r in csv.DictReader(l for l in open(pre + "_isa_histogram.csv") if not l.startswith("#")):
        is_kernel = "k_stage" in r["symbol"]
        if (is_kernel and r["scope"] == "loops") or (not is_kernel and r["scope"] == "text"):
            if r["opcode"].startswith("v_"):
                hist[r["kernel"]][r["opcode"]] += int(r["count"])
    kernels = {}
    occupancy = load_occupancy(pre + "_kernel_meta.txt")
    print("%-24s %12s %7s %7s %7s %7s %9s" % ("kernel", "SQ_INSTS_VALU", "mad", "vop3", "simple", "carry", "cyc/inst"))
    for k, stage in STAGE_OF.items():
        if k not in pmc or "SQ_INSTS_VALU" not in pmc[k]:
            continue
        c = pmc[k]
        # waves per SIMD the kernel's registers allow (the big kernels: one; the fixed-base chunks since round 5: two)
        waves_per_simd = min(2, occupancy.get(k, 1))
        cost = class_costs(ub, waves_per_simd)
        own = bool(hist.get(k))
        h = hist.get(k) or hist.get("PvT1Chain<BlsCurve>") or hist.get("PvMsmPart<BlsCurve>")      # no histogram of its own: a chain kernel's mix
        tot = sum(h.values())
        share = collections.Counter()
        for op, n in h.items():
            share[classify(op)] += n / tot
        def cpi_at(w):
            cw = class_costs(ub, w)
            return sum(share[cl] * cw[cl] for cl in share)
        def ns_at(w):
            cw = class_costs_ns(ub, w)
            return sum(share[cl] * cw[cl] for cl in share)
        cpi = cpi_at(waves_per_simd)
        # clock = GRBM_GUI_ACTIVE / 8 / duration: only meaningful for a kernel that runs long enough for the counter window to
        # be the kernel (a 10 us kernel reads 9 GHz)
        clk = c.get("GRBM_GUI_ACTIVE", 0) / 8.0 / c["duration_ns"] if c.get("duration_ns", 0) > 500000 else None
        kernels[stage] = {"kernel": k, "valu_insts": c["SQ_INSTS_VALU"], "FETCH_SIZE_KiB": c.get("FETCH_SIZE", 0.0), "WRITE_SIZE_KiB": c.get("WRITE_SIZE", 0.0),
                          "waves": c.get("SQ_WAVES"), "waves_per_simd": waves_per_simd, "cycles_per_inst": cpi, "own_isa_histogram": own,
                          "cycles_per_inst_if_waves_per_simd": {str(w): round(cpi_at(w), 3) for w in (1, 2, 4, 8)},
                          "ns_per_inst_if_waves_per_simd": {str(w): round(ns_at(w), 4) for w in (1, 2, 4, 8)},
                          "opcode_class_share": {cl: round(share[cl], 4) for cl in ("mad", "vop3", "simple", "carry")},
                          "class_cost_cycles": {cl: round(cost[cl], 3) for cl in cost},
                          "wait_any_over_wave_cycles": c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else None,
                          "duration_ns_exclusive": c.get("duration_ns"), "clock_ghz_grbm": clk}
        print("%-24s %12.4g %7.3f %7.3f %7.3f %7.3f %9.3f" % (k, c["SQ_INSTS_VALU"], share["mad"], share["vop3"], share["simple"], share["carry"], cpi))
    big = [v for v in kernels.values() if v["clock_ghz_grbm"] and v["valu_insts"] > 1e8]
    clock = sum(v["clock_ghz_grbm"] * v["duration_ns_exclusive"] for v in big) / sum(v["duration_ns_exclusive"] for v in big)
    # the sources the profiled library was built from: bench.py compares this with the library it runs and flags a mismatch
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bbs_sign_amd import build as _build
    ops = model_ops(pre, ub, hist, occupancy, classify)
    out = {"kernels": kernels, "ops": ops, "clock_ghz_under_load": clock, "library_source_hash": os.environ.get("BBS_PROFILED_HASH") or _build.source_hash(),
           "source": "%s_{pmc,ubench_valu_int,isa_histogram}.csv via tools/valu_model.py" % pre,
           "note": "SQ_INSTS_VALU / FETCH_SIZE / WRITE_SIZE: rocprofv3 --pmc, mean of the last launches, one 4096-item BLS12-381 batch; rocprofv3 "
                   "serialises dispatches while collecting counters, so cycle counters are exclusive-run values; clock = GRBM_GUI_ACTIVE / 8 / kernel duration"}
    with open(pre + "_counters.json", "w") as f:
        json.dump(out, f, indent=1)
    print("clock under load (GRBM_GUI_ACTIVE / 8 / duration, big kernels): %.3f GHz" % clock)
    print("wrote", pre + "_counters.json")


if __name__ == "__main__":
    main()
