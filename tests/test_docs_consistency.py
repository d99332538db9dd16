"""Documentation drift guards (CPU): every PYLAMP_* environment variable the product reads is listed in README.md."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_environment_knob_is_documented():
    readme = open(os.path.join(ROOT, "README.md")).read()
    names = set()
    for pat in ("pylamp_amd/csrc/*.hip", "pylamp_amd/csrc/*.h", "pylamp_amd/*.py", "bench.py"):
        for f in glob.glob(os.path.join(ROOT, pat)):
            names |= set(re.findall(r"PYLAMP_[A-Z0-9_]+", open(f).read()))
    names -= {"PYLAMP_HIP_H", "PYLAMP_"}                    # (include guard, prefix fragments)
    missing = sorted(n for n in names if n not in readme)
    assert not missing, "README.md does not list: %s" % missing
