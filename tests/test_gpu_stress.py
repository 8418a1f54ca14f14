"""-m gpu: a short run of tools/stress_schedules.py inside the suite -- random tile shapes (ragged, unaligned, 1 x w), batch
sizes on both sides of the schedule switch, contents with heavy ties / mostly background / empty tiles, random extractor
parameters: the fused kernel and the one-launch-per-phase schedule must agree (Macenko to the byte, Vahadane to the
summation order)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed,cases,method", [(101, 30, "macenko"), (102, 20, "vahadane")])
def test_random_cases_agree_across_schedules(seed, cases, method):
    r = subprocess.run([sys.executable, os.path.join("tools", "stress_schedules.py"), str(seed), str(cases), method],
                       cwd=REPO, capture_output=True, text=True, timeout=900)
    tail = "\n".join(r.stdout.strip().splitlines()[-5:])
    assert r.returncode == 0, tail + r.stderr[-2000:]
    assert "mismatching cases: 0" in r.stdout, tail
    assert r.stdout.count(" OK") >= cases
