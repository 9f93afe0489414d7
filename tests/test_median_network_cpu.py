"""CPU: the 99-comparator selection network of k_median5_u16 (sindslam_amd/csrc/median25_net.inc) returns the median of 25 values.
0-1 principle: a comparator network that leaves the median of EVERY 0/1 input in p[12] does so for every input -- all 2^25 of them are checked."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include <stdio.h>
#include <stdint.h>
int main(void) {
    long bad = 0; int ncmp = 0;
#define S(a, b) ncmp++;
#include "median25_net.inc"
#undef S
    for (uint32_t m = 0; m < (1u << 25); m++) {
        int p[25];
        for (int i = 0; i < 25; i++) p[i] = (m >> i) & 1;
#define S(a, b) { const int lo = p[a] < p[b] ? p[a] : p[b], hi = p[a] < p[b] ? p[b] : p[a]; p[a] = lo; p[b] = hi; }
#include "median25_net.inc"
#undef S
        if (p[12] != (__builtin_popcount(m) >= 13)) bad++;
    }
    printf("%d %ld\n", ncmp, bad);
    return 0;
}
'''


def test_median25_network_selects_the_median_of_every_binary_input():
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(SRC)
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-O2", "-I", os.path.join(ROOT, "sindslam_amd", "csrc"), "-o", exe, c])
        ncmp, bad = subprocess.check_output([exe], timeout=300).decode().split()
    assert int(ncmp) == 99 and int(bad) == 0
