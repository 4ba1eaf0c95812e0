#!/usr/bin/env python3
"""Rewrites profiles/traffic.json (bench.py's fallback when rocprofv3 is unavailable) from a bench
line whose roofline objects carry live PMC traffic, stamped with the hash of the CURRENT
fm_index_amd/csrc sources:   python profiles/update_traffic.py profiles/r02/bench_default.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fm_index_amd import _lib as L   # noqa: E402


def ent(r, src):
    assert r.get("traffic") and r.get("fetch_kb_raw"), "the bench line has no live PMC traffic"
    return {"fetch_kb_raw": r["fetch_kb_raw"], "write_kb": r["write_kb"],
            "kernel": r.get("traffic_kernel"), "source": src + " (live rocprofv3 --pmc passes of that run)"}


def main():
    path = sys.argv[1]
    d = json.loads(open(path).read().strip().splitlines()[-1])
    src = os.path.relpath(os.path.abspath(path), ROOT)
    h = L.csrc_hash()
    cfg = d["config"]
    key = "dna:%d:%d:%d" % (cfg["patterns_per_gpu"], cfg["pattern_len"], cfg["text_len"].bit_length() - 1)
    t = {"_comment": "Fallback for bench.py when rocprofv3 is not available: raw FETCH_SIZE / WRITE_SIZE per launch from the "
                     "live PMC passes of the named run (bench.py prices them: price_traffic).  An entry is only used "
                     "when `csrc_hash` equals the hash of the current fm_index_amd/csrc sources "
                     "(fm_index_amd/_lib.py::csrc_hash); regenerate with profiles/update_traffic.py.",
         key: {"csrc_hash": h, "count": ent(d["roofline"], src), "locate": ent(d["locate"]["roofline"], src)}}
    r3b = (d.get("locate_3b") or {}).get("roofline")
    if r3b and r3b.get("traffic") and r3b.get("fetch_kb_raw"):
        t[key]["locate_3b"] = ent(r3b, src)
    if "rlfm" in d and "roofline" in d["rlfm"]:
        rc = d["rlfm"]["config"]
        rkey = "bytes-rlfm:%d:%d:%d" % (rc["patterns"], rc["pattern_len"], rc["text_len"].bit_length() - 1)
        t[rkey] = {"csrc_hash": h, "count": ent(d["rlfm"]["roofline"], src),
                   "locate": ent(d["rlfm"]["locate"]["roofline"], src)}
    json.dump(t, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    print("profiles/traffic.json <-", src, "csrc_hash", h)


if __name__ == "__main__":
    main()
