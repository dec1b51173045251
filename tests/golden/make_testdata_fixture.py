"""Convert the reference's test-data (GloVe-50d text: `word v1 .. v50`) into binary fixtures.

Run once in the build container (needs /root/reference, which does not exist on the GPU box):
    python tests/golden/make_testdata_fixture.py
Writes tests/golden/testdata_store.npy (1000x50 f32) and testdata_queries.npy (100x50 f32).
These are DATA (inputs the reference's own test hnsw_glove_build_eval uses,
hnsw/src/template.rs:518-523), not code.

Parsing follows load_glove_array (hnsw/src/helpers/glove.rs:38-69): split on ' ', first token is
the word, every later token that parses as f32 is a value, any other token is appended to the
word.  Rust's str::parse::<f32> is correctly rounded from the decimal string; going through a
Python double first could double-round, so each value is rounded exactly with Fractions.
"""
import os
import sys
from fractions import Fraction

import numpy as np

REF = "/root/reference/test-data"
HERE = os.path.dirname(os.path.abspath(__file__))


def parse_f32_exact(tok):
    try:
        x = float(tok)
    except ValueError:
        return None
    if x != x or x in (float("inf"), float("-inf")):
        return np.float32(x)
    exact = Fraction(tok)
    c = np.float32(x)
    best = c
    for cand in (np.nextafter(c, np.float32(-np.inf)), np.nextafter(c, np.float32(np.inf))):
        if not np.isfinite(cand):
            continue
        ea, eb = abs(Fraction(float(cand)) - exact), abs(Fraction(float(best)) - exact)
        if ea < eb or (ea == eb and (int(np.float32(cand).view(np.uint32)) & 1) == 0):
            best = cand
    return np.float32(best)


def load(path):
    rows, words = [], []
    with open(path) as f:
        for line in f:
            parts = line.rstrip("\n").split(" ")
            word, vals = parts[0], []
            for tok in parts[1:]:
                v = parse_f32_exact(tok)
                if v is None:
                    word += tok
                else:
                    vals.append(v)
            rows.append(vals)
            words.append(word)
    return words, np.array(rows, dtype=np.float32)


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference test-data not present; fixtures are already committed")
    for name in ("store", "queries"):
        words, arr = load(os.path.join(REF, name + ".txt"))
        print(name, arr.shape, arr.dtype, "mean %.4f std %.4f" % (arr.mean(), arr.std()))
        np.save(os.path.join(HERE, "testdata_%s.npy" % name), arr)
