// Host-side check of the XCD-aware tile walk (shift-net_amd/csrc/sn_common.h): for many grid shapes every (t, ty, tx) must be
// produced by exactly one workgroup of the launch grid, padding workgroups must be rejected, and the multiply-high division must
// agree with the plain one.  Built and run by tests/test_host_logic.py with hipcc in host-only mode (no GPU needed).
#include "sn_common.h"
#include <cstdio>
#include <vector>

int main() {
    long checked = 0;
    const int ntxs[] = {1, 2, 3, 7, 8, 20, 40, 60, 135};
    const int ntys[] = {1, 2, 3, 5, 23, 45, 90, 135, 270};
    const int Ts[] = {1, 2, 5, 8, 9, 20, 36, 52, 100};
    for (int ntx : ntxs) for (int nty : ntys) for (int T : Ts) {
        const XcdTiles g = sn_xcd_tiles(ntx, nty, T);
        const dim3 grid = sn_xcd_grid(g);
        std::vector<unsigned char> seen((size_t)ntx * nty * T, 0);
        long live = 0;
        for (uint32_t by = 0; by < grid.y; ++by) for (uint32_t bx = 0; bx < grid.x; ++bx) {
            int t, ty, tx;
            if (!sn_xcd_decode(g, bx, by, t, ty, tx)) continue;
            if (t < 0 || t >= T || ty < 0 || ty >= nty || tx < 0 || tx >= ntx) { std::printf("out of range: ntx %d nty %d T %d -> (%d, %d, %d)\n", ntx, nty, T, t, ty, tx); return 1; }
            unsigned char& s = seen[((size_t)t * nty + ty) * ntx + tx];
            if (s) { std::printf("duplicate tile: ntx %d nty %d T %d -> (%d, %d, %d)\n", ntx, nty, T, t, ty, tx); return 1; }
            s = 1; ++live;
        }
        if (live != (long)ntx * nty * T) { std::printf("missing tiles: ntx %d nty %d T %d: %ld of %ld\n", ntx, nty, T, live, (long)ntx * nty * T); return 1; }
        if ((long)grid.x * grid.y - live >= 8L * ntx + (SN_XCD_TILES ? 0 : 1) && SN_XCD_TILES) { std::printf("too much padding: ntx %d nty %d T %d\n", ntx, nty, T); return 1; }
        checked += live;
    }
    std::printf("xcd tiles ok: %ld tiles over %zu grid shapes\n", checked, sizeof(ntxs) / sizeof(int) * sizeof(ntys) / sizeof(int) * sizeof(Ts) / sizeof(int));
    return 0;
}
