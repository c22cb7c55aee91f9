"""-m gpu: a short soak of the drop-in path (tools/soak_mirror.py: dabgpu_radio_cli fed a raw_u8 capture through a pipe, the framing broken at every seam):
6,000 frames with 59 losses of lock and re-acquisitions -- frames, FIB CRCs and sub-channel bytes complete, and neither the process nor the device grows.
(A stream that only carried copies and event records and was never synchronised once cost 2.4 KB of host memory per frame: profiles/r05/ab_notes.md §9.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_growth_over_six_thousand_frames_and_sixty_reacquisitions():
    cli = os.path.join(ROOT, "dab-radio_amd", "host", "apps", "dabgpu_radio_cli")
    if not os.path.exists(cli):
        import __graft_entry__ as g
        g.build()
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak_mirror.py"), "--frames", "100", "--repeats", "60"], capture_output=True, text=True, timeout=600)
    out = json.loads(res.stdout.strip().splitlines()[-1]) if res.stdout.strip() else {}
    assert res.returncode == 0 and out.get("ok"), (res.stderr[-1500:], out)
    assert out["frames_read"] >= 60 * 98 and out["frames_desync"] >= 55
    assert out["memory_at_end"]["host_rss_MB"] <= out["memory_at_25_percent"]["host_rss_MB"] + 8
