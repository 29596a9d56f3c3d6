#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (rocpd sqlite DBs, one counter group per DB) into one JSON:
per kernel, per counter, the mean value per dispatch.  HBM traffic is derived as the guide
(/opt/skills/guides/MI355X_MICROARCH.md, section HBM) prescribes: FETCH_SIZE and WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE reports exactly half of the bytes of wide (16 B/lane) coalesced reads, so the
read side is doubled; WRITE_SIZE is used as reported (uncalibrated).
Usage: python tools/pmc_summary.py gpurun_out/pmc/*.db > profiles/rNN_pmc_summary.json"""
import json
import sqlite3
import sys


def main(paths):
    out = {}
    for path in paths:
        cur = sqlite3.connect(path).cursor()
        q = ("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) "
             "from counters_collection group by kernel_name, counter_name")
        rows = list(cur.execute(q))
        lds_pass = any(ctr == "SQ_LDS_IDX_ACTIVE" for _k, ctr, _t, _n in rows)
        for kern, ctr, total, ndisp in rows:
            if lds_pass and ctr == "GRBM_GUI_ACTIVE":
                ctr = "GRBM_GUI_ACTIVE_LDS"          # (the MFMA pass has its own GRBM_GUI_ACTIVE)
            out.setdefault(kern, {})[ctr] = total / max(1, ndisp)
            out[kern]["dispatches_" + ctr] = ndisp
    for kern, c in out.items():
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            c["hbm_read_bytes_corrected"] = c["FETCH_SIZE"] * 1024 * 2
            c["hbm_write_bytes"] = c["WRITE_SIZE"] * 1024
            c["hbm_bytes_per_launch"] = c["hbm_read_bytes_corrected"] + c["hbm_write_bytes"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs on the chip
            c["mfma_util"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0)
        if "SQ_LDS_IDX_ACTIVE" in c and "SQ_LDS_BANK_CONFLICT" in c:
            # share of the LDS's busy cycles lost to bank conflicts, and (with GRBM_GUI_ACTIVE of the same pass) how busy the
            # 256 LDS units were over the kernel
            c["lds_bank_conflict_share"] = c["SQ_LDS_BANK_CONFLICT"] / max(1.0, c["SQ_LDS_IDX_ACTIVE"])
            if "GRBM_GUI_ACTIVE_LDS" in c:
                c["lds_util"] = c["SQ_LDS_IDX_ACTIVE"] / (256.0 * c["GRBM_GUI_ACTIVE_LDS"] / 8.0)
    # which sources the passes ran on: POCR_SOURCE_HEAD (the evidence script exports `git rev-parse HEAD` of the build container;
    # the GPU box has no .git) and the source hash of the library that ran (__graft_entry__.built_hash)
    import os
    meta = {"head": os.environ.get("POCR_SOURCE_HEAD", ""), "library_source_hash": ""}
    try:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import __graft_entry__ as ge
        meta["library_source_hash"] = ge.built_hash()
    except Exception:
        pass
    out["_meta"] = meta
    json.dump(out, sys.stdout, indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1:])
