"""Collects the reference's own golden vectors for the import/diff_calc path into one JSON file.

Run in the build container (needs /root/reference):  python tests/golden/make_reference_fixtures.py
Sources (paths relative to /root/reference):
  loro-js/tests/fixtures/rust/*.blob + *.json   consumed by crates/loro/tests/loro_js_interop.rs:42-126
The blobs are stored hex-encoded with their expected deep JSON so that nothing under tests/ has to read
/root/reference at run time (it does not exist on the GPU box).
"""
import json, os, sys

SRC = "/root/reference/loro-js/tests/fixtures/rust"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_fixtures.json")

BLOBS = ["updates.blob", "updates.ts.blob", "concurrent-base.ts.blob", "concurrent-left.ts.blob",
         "concurrent-right.ts.blob", "fugue-left.ts.blob", "fugue-right.ts.blob", "runtime-updates.ts.blob",
         # FastSnapshot (mode 3) fixtures: the same histories as updates.blob / runtime-updates.ts.blob, and a shallow one
         "snapshot.blob", "snapshot.ts.blob", "runtime-snapshot.ts.blob", "shallow.ts.blob"]
JSONS = ["snapshot.deep.json", "concurrent.expected.json", "runtime.expected.json", "meta.json"]


def main():
    out = {"_provenance": "loro-dev/loro loro-js/tests/fixtures/rust (see make_reference_fixtures.py)", "blobs": {}, "json": {}}
    for b in BLOBS:
        out["blobs"][b] = open(os.path.join(SRC, b), "rb").read().hex()
    for j in JSONS:
        out["json"][j] = json.load(open(os.path.join(SRC, j)))
    json.dump(out, open(OUT, "w"), indent=1, sort_keys=True)
    print("wrote", OUT)


if __name__ == "__main__":
    sys.exit(main())
