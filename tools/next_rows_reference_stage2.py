#!/usr/bin/env python3
"""tools/next_rows_reference_stage2.py -- bench.py's `next_rows` command-line leg with the compiled reference's stage II actually RUN on the
same sketch directory (minutes: it builds and writes the dense 32 GiB index whatever the input) and the files compared -- what the default
bench points to instead of repeating.  Prints the leg's JSON (-> profiles/r06_next_rows_reference_stage2.json).

    python3 tools/next_rows_reference_stage2.py [--timeout 900]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--timeout", type=int, default=900)
    a = ap.parse_args()
    import bench
    from metakssd_amd import capi
    print(json.dumps(bench.leg_next_rows_cli(capi, ref_stage2_timeout=a.timeout)))


if __name__ == "__main__":
    main()
