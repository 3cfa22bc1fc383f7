"""Summarise rocprofv3 --pmc CSVs: mean counter value per kernel name; with --traffic also write
profiles-style JSON {kernel: {"hbm_bytes_per_launch": ...}} using the gfx950 correction of
MI355X_MICROARCH.md (FETCH_SIZE reports half of a wide coalesced read; units are KiB)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "?")
            if "d4w::" not in k:
                continue
            k = k.split("(")[0].replace("void d4w::", "").replace("d4w::", "")
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("   %-32s %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
if "--traffic" in sys.argv:
    alias = {"fk_passA_fwd": "fk_passA_fwd", "fk_passA_inv": "fk_passA_inv", "fk_passB": "fk_passB_mid"}
    out = {}
    for k, cs in acc.items():
        if "FETCH_SIZE" not in cs or "WRITE_SIZE" not in cs:
            continue
        fetch = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"]) * 1024.0
        write = sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"]) * 1024.0
        name = k.split("<")[0]
        if name in ("fk_passC", "fkf_passC"):
            # generic kernel: fk_passC<INV, GENERIC>; specialised: fkf_passC<FkFastCfg<...>, INV, MODE>
            targ = k[k.rindex(">, ") + 3:] if name == "fkf_passC" and ">, " in k else k[k.index("<") + 1:]
            name = "fk_passC_inv" if targ.strip().startswith("true") else "fk_passC_fwd"
        if name == "fkf_passBt":                        # time-first order: fkf_passBt<FkFastCfg<...>, PHASE>
            targ = k[k.rindex(">, ") + 3:] if ">, " in k else "1"
            name = "fk_passBi_time_inv" if targ.strip().startswith("2") else "fk_passBf_time_fwd"
        name = {"fkf_passA_fwd": "fk_passA_fwd", "fkf_passA_inv": "fk_passA_inv", "fkf_passA_inv_stats": "fk_passA_inv",
                "fkf_passB": "fk_passB", "fkf_passCm": "fk_passCm_channel"}.get(name, name)
        if (name == "xcorr_fft_blocks" and "<1, true" in k) or name == "xcorr_fft_fused4":
            name = "xcorr_fft_fused"                    # the two-template launch (one read, two correlograms)
        name = alias.get(name, name)
        if name == "xcorr_mm_rows":
            # xcorr_mm_rows<KS0, KS1, WPS, TAIL, WMAX>: the kernel with the zero-padded template's tail (what the public step
            # launches, round 6) under the plain name, the one without it beside it
            targs = [t.strip() for t in k[k.index("<") + 1:k.rindex(">")].split(",")]
            if not (len(targs) >= 4 and targs[3] == "true"):
                name = "xcorr_mm_rows_no_tail"
        # gfx950: FETCH_SIZE tallies a 128-byte request as 64 bytes (MI355X_MICROARCH.md, HBM), so kernels whose reads
        # are >= 128-byte contiguous per row piece are doubled: every pass of the shape-specialised f-k kernels reads
        # 128-byte strips (TA = TC = 16 complex).  Only the GENERIC pass C at TC = 8 (64-byte strips) is counted in
        # full -- calibrated on the known byte count of the block (raw FETCH_SIZE == 9.6 GB == the block).
        factor = 1.0 if (name.startswith("fk_passC") and k.startswith("fk_passC")) else 2.0
        out[name] = {"kernel": k, "fetch_size_bytes_raw": fetch, "write_size_bytes": write,
                     "fetch_factor": factor, "hbm_bytes_per_launch": factor * fetch + write,
                     "note": "FETCH_SIZE in KiB; x2 for wide (>=128 B) requests, x1 for 64-byte strips"}
    # where the numbers come from (bench.py prints it as roofline.traffic_source): PMC_SOURCE, or the directory summarised
    out["_source"] = os.environ.get("PMC_SOURCE", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes summarised from %s (scripts/pmc.sh)" % root)
    json.dump(out, open(sys.argv[sys.argv.index("--traffic") + 1], "w"), indent=1, sort_keys=True)
