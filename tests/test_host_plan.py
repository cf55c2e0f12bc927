"""Host-side plans of round 5 (mri_inr_amd/csrc/host_plan.h, plain C++ compiled with g++): the chunk plan of a synchronous host-pointer call
that pipelines itself, and the section layout / ring-depth rule of the one-launch prologue's packed weight stream."""
import os
import shutil
import subprocess
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROG = textwrap.dedent(r"""
    #include <cassert>
    #include <cstdio>
    #include "host_plan.h"
    using namespace msiren;

    static void check_plan(int64_t B, int64_t first, int64_t piece, int s0) {
        const auto plan = pipelined_host_plan(B, first, piece, s0);
        assert(!plan.empty());
        int64_t at = 0;
        for (size_t k = 0; k < plan.size(); ++k) {
            const auto& c = plan[k];
            assert(c.lo == at && c.n > 0);                       // contiguous, in order, nothing empty
            at += c.n;
            assert(c.stream == ((int)(k & 1) ^ (s0 & 1)));       // chunks alternate between the two streams
            assert(c.beside == (k > 0));                         // only chunk 0's prologue has the chip to itself
            assert(c.trunk == (k + 1 == plan.size() ? 2 : 1));   // the last chunk takes the weight-stationary trunk, every other one leaves room
            if (k > 0 && k + 1 < plan.size()) assert(c.n == piece);
            if (k > 0 && k + 1 == plan.size()) assert(c.n >= std::min<int64_t>(128, B - plan[0].n) && c.n < piece + 128);
        }
        assert(at == B);                                          // every tile exactly once
        assert(plan[0].n <= std::max<int64_t>(16, B / 3) || plan[0].n == B);
    }

    int main() {
        for (int64_t B : {1, 15, 16, 127, 128, 400, 799, 800, 801, 1000, 1339, 3200, 3300, 25600, 25601})
            for (int64_t first : {16, 56, 112, 400, 5000})
                for (int64_t piece : {64, 144, 400, 1600})
                    for (int s0 : {0, 1}) check_plan(B, first, piece, s0);
        {   // the shipped knobs at one slice pair and at 8 slices
            const auto p = pipelined_host_plan(800, 112, 400, 0);
            assert(p.size() == 3 && p[0].n == 112 && p[1].n == 400 && p[2].n == 288);
            const auto q = pipelined_host_plan(3200, 112, 400, 1);
            assert(q.size() == 9 && q[0].stream == 1 && q[8].n == 288 && q[8].trunk == 2);
            const auto r = pipelined_host_plan(912, 112, 400, 0);   // 400 left behind the second chunk: not below 128 -> its own chunk
            assert(r.size() == 3 && r[2].n == 400);
            const auto t = pipelined_host_plan(1000, 112, 400, 0);  // 88 left: absorbed by the last chunk
            assert(t.size() == 3 && t[2].n == 488);
        }
        // the packed weight stream: H = Z = 256, L = 5 (every shipped YAML): 32 + 8 + 80 + 64 k-steps per wave = 2.9 MB in all
        {
            const auto s = em_stream_layout(2, 2, 5, true, true);
            assert(s.c3 == 32 && s.fc == 8 && s.zp == 80 && s.hl == 64 && s.zp_start == 40 && s.hl_start == 120 && s.total == 184);
            assert((long)s.total * 4 /*waves*/ * 4096 == 3014656);
            assert(s.z_pass(0, 0, 2, 8) == 40 && s.z_pass(4, 1, 2, 8) == 112 && s.h_pass(1, 0, 2, 8) == 120 && s.h_pass(4, 1, 2, 8) == 176);
            const auto m = em_stream_layout(2, 2, 5, false, true);   // a trunk + Modulator checkpoint: the stream starts at the latent stage
            assert(m.zp_start == 0 && m.total == 144);
            const auto e = em_stream_layout(2, 2, 5, true, false);   // encoder only
            assert(e.total == 40 && e.zp == 0 && e.hl == 0);
            const auto c5 = em_stream_layout(4, 1, 10, true, true);  // config 5: H = 512, Z = 128, L = 10
            assert(c5.c3 == 32 && c5.fc == 4 && c5.zp == 160 && c5.hl == 576 && c5.total == 772);
            const auto l1 = em_stream_layout(2, 2, 1, true, true);   // a single layer has no hidden chain
            assert(l1.hl == 0 && l1.total == 32 + 8 + 16);
        }
        // ring depths: every instantiated (NPH, NPZ, DEPTH) passes; the ones that would need run-time slots do not
        for (int d : {2, 4, 8}) assert(em_ring_depth_ok(2, 2, d));
        assert(em_ring_depth_ok(2, 2, 16));                          // (legal; not shipped: the compiler spills 140-200 registers)
        assert(em_ring_depth_ok(4, 1, 4) && em_ring_depth_ok(4, 1, 8) && em_ring_depth_ok(4, 1, 16));
        assert(!em_ring_depth_ok(2, 2, 3) && !em_ring_depth_ok(2, 2, 32) && !em_ring_depth_ok(4, 1, 32) && !em_ring_depth_ok(2, 2, 0));
        std::puts("ok");
        return 0;
    }
""")


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not installed")
def test_host_plans(tmp_path):
    src = tmp_path / "plan.cpp"
    src.write_text(PROG)
    exe = tmp_path / "plan"
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "mri_inr_amd", "csrc"), str(src), "-o", str(exe)],
                   check=True, capture_output=True, text=True)
    res = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert res.returncode == 0 and res.stdout.strip() == "ok", res.stderr
