#!/usr/bin/env python3
"""Golden vectors for the segment cut search: the REAL reference BeamSearch
(speechcatcher/simple_endpointing.py:21-79), imported read-only with import
shims for the two absent third-party modules, run on seeded energy curves.
Run in the survey container only:  python tools/gen_golden_segmenter.py"""
import json
import os
import sys
from pathlib import Path

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools" / "ref_shim"))
sys.path.insert(0, "/root/reference/speechcatcher")

from speechcatcher_amd.synth import synth_energy_curve  # noqa: E402

import simple_endpointing as ref  # noqa: E402


def main():
    cases = []
    for seed, n, avg, params in [
        (1, 9000, 30.0, dict(beam_size=10, step=10, len_reward_weight=12.0, energy_weight=1.0)),
        (2, 20000, 60.0, dict(beam_size=10, step=10, len_reward_weight=12.0, energy_weight=1.0)),
        (3, 12000, 20.0, dict(beam_size=4, step=25, len_reward_weight=1.0, energy_weight=1.0)),
        (4, 3000, 60.0, dict(beam_size=10, step=10, len_reward_weight=12.0, energy_weight=1.0)),
        (5, 26000, 40.0, dict(beam_size=6, step=10, len_reward_weight=6.0, energy_weight=2.0)),
    ]:
        bs = ref.BeamSearch(ideal_segment_len=int(avg * 100), **params)
        e = synth_energy_curve(seed, n)
        segs = bs.search(e, n)
        cases.append({"seed": seed, "n": n, "average_segment_length": avg, "params": params,
                      "segments": [[int(a), int(b)] for a, b in segs]})
        print(seed, n, len(segs), segs[:3])
    out = ROOT / "tests" / "golden" / "segmenter.json"
    out.write_text(json.dumps({"cases": cases}, indent=1))
    print("wrote", out)


if __name__ == "__main__":
    main()
