"""Host bookkeeping of the persistent trunks' pass queue (mri_inr_amd/csrc/pass_queue.h), compiled with g++ and
driven against a simulated device counter: the host value moves only after an accepted launch, so a failed
launch can never leave it ahead of the device (which would hand out negative pass ids)."""
import os
import shutil
import subprocess
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROG = textwrap.dedent(r"""
    #include <cassert>
    #include <cstdio>
    #include <set>
    #include <vector>
    #include "pass_queue.h"
    using msiren::PassQueue;

    // the device side of one launch: `grid` workgroups, workgroup g starts with pass g, every executed pass does one
    // atomicAdd and takes next_pass(); returns the set of executed passes
    static std::multiset<int> run_launch(unsigned& counter, unsigned base, int grid, int npasses) {
        std::multiset<int> done;
        std::vector<int> cur(grid);
        for (int g = 0; g < grid; ++g) cur[g] = g;
        bool any = true;
        while (any) {
            any = false;
            for (int g = 0; g < grid; ++g)
                if ((unsigned)cur[g] < (unsigned)npasses) {  // the kernels' unsigned guard
                    done.insert(cur[g]);
                    const unsigned old = counter++;
                    cur[g] = PassQueue::next_pass(old, base, grid);
                    any = true;
                }
        }
        return done;
    }

    static void expect_all(const std::multiset<int>& done, int npasses) {
        assert((int)done.size() == npasses);
        int want = 0;
        for (int p : done) assert(p == want++);
    }

    int main() {
        for (unsigned start : {0u, 0xfffffff0u, 0x7ffffff8u}) {  // also across the 2^32 and 2^31 wrap
            unsigned counter = start;
            PassQueue q;
            q.reset(start);
            for (int it = 0; it < 50; ++it) {
                const int npasses = 1 + (it * 37) % 1900, grid = npasses < 256 ? npasses : 256;
                const unsigned base = q.begin(npasses);
                if (it % 7 == 3) {  // launch refused (attribute / launch error): nothing ran on the device
                    q.abort();
                    assert(q.base == counter);
                    continue;
                }
                expect_all(run_launch(counter, base, grid, npasses), npasses);
                q.commit();
                assert(q.base == counter);  // host and device agree after every accepted launch
            }
            // a launch whose pass count only the device knows, followed by the reset
            (void)q.begin(123);
            q.commit();
            counter = 0;
            q.reset(0);
            expect_all(run_launch(counter, q.begin(10), 4, 10), 10);
            q.commit();
            assert(q.base == counter);
        }
        // what the old bookkeeping did on a failed launch (advance first): ids go negative and the guard stops them
        {
            unsigned counter = 0;
            const unsigned stale_base = 500;  // host 500 ahead of the device
            const auto done = run_launch(counter, stale_base, 4, 100);
            assert(done.size() == 4);  // only the initial passes; every fetched id is negative -> workgroup ends
        }
        std::puts("pass queue ok");
        return 0;
    }
""")


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not installed")
def test_pass_queue_bookkeeping(tmp_path):
    src = tmp_path / "pq.cpp"
    src.write_text(PROG)
    exe = tmp_path / "pq"
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "mri_inr_amd", "csrc"), str(src), "-o", str(exe)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "pass queue ok" in r.stdout, r.stdout + r.stderr
