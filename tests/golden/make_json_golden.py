#!/usr/bin/env python3
"""Generate tests/golden/model_json.json.gz by RUNNING THE REFERENCE's pomegranate: HiddenMarkovModel.to_json /
from_json (hmm.pyx:3023-3143), the format of the stored HMMs of vntr_finder.py:124-137, on a read-matcher model and on
the repeat finder.  TEST INFRASTRUCTURE; only data is written (the JSON text the reference wrote and the baked model it
loads back from it).

    python oracle/tools/build_reference.py && python tests/golden/make_json_golden.py
"""
import gzip
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF_BUILD = os.environ.get("ADVNTR_REF_BUILD", "/tmp/advntr_ref_build")
sys.path[:0] = [os.path.join(REPO, "oracle", "tools", "nx111"), os.path.join(REPO, "oracle", "tools", "stubs"), REF_BUILD, REPO]

import numpy as np                                    # noqa: E402
from advntr import settings, hmm_utils                # noqa: E402
from pomegranate import HiddenMarkovModel             # noqa: E402
from make_merge_golden import dump                    # noqa: E402


def main():
    rng = np.random.default_rng(31)
    dna = lambda n: "".join("ACGT"[i] for i in rng.integers(0, 4, n))
    settings.MAX_ERROR_RATE = 0.05
    cases = []
    left, right, pattern = dna(12), dna(10), dna(7)
    m1 = hmm_utils.get_read_matcher_model(left, right, [pattern], 2)
    m2 = hmm_utils.build_reference_repeat_finder_hmm(["ACGTT"], copies=2)
    for name, m, spec in (("read_matcher", m1, {"left": left, "right": right, "pattern": pattern, "copies": 2}),
                          ("repeat_finder", m2, {"pattern": "ACGTT", "copies": 2})):
        text = m.to_json()
        loaded = HiddenMarkovModel.from_json(text)
        reads = [dna(int(rng.integers(5, 30))) for _ in range(5)]
        scores = [loaded.viterbi(r)[0] for r in reads]
        cases.append({"name": name, "spec": spec, "json": text, "loaded": dump(loaded), "reads": reads, "logp": scores})
        print(name, "states", len(m.states), "->", len(loaded.states), "json bytes", len(text))
    # cross-check at generation time: JSON written by the product's mirror loads into the reference to the same model
    from advntr_amd import hmm_utils as mine, settings as my_settings
    my_settings.MAX_ERROR_RATE = 0.05
    from oracle import stepwise_builder
    mm = stepwise_builder.get_read_matcher_model(left, right, [pattern], 2)
    ref_from_mine = HiddenMarkovModel.from_json(mm.to_json())
    assert dump(ref_from_mine)["state_names"] == cases[0]["loaded"]["state_names"]
    a, b = dump(ref_from_mine)["edges"], cases[0]["loaded"]["edges"]
    assert [(x[0], x[1]) for x in a] == [(x[0], x[1]) for x in b]
    assert max(abs(x[2] - y[2]) for x, y in zip(a, b) if np.isfinite(y[2])) < 1e-12
    print("reference.from_json(product.to_json()) == reference.from_json(reference.to_json())")
    with gzip.open(os.path.join(HERE, "model_json.json.gz"), "wt") as fh:
        json.dump({"cases": cases}, fh)


if __name__ == "__main__":
    sys.path.insert(0, HERE)
    main()
