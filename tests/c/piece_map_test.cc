// piece_map_test.cc -- the reconstruction kernels' workgroup -> (item, row group, segment) arithmetic
// (amv-codec-tools_amd/csrc/amv_piece_map.h) checked on the CPU: for every shape, every workgroup number of a launch
// names a piece exactly once, by the stated formula, and the divisions by reciprocal agree with real divisions right up
// to the launch size the host allows.  Built with g++ by tests/test_abi_and_host.py; prints "ok <cases>" or the failure.
#include <cstdio>
#include <vector>

#include "amv_piece_map.h"

using namespace amv;

static int check_launch(uint32_t row_groups, uint32_t nseg, uint32_t walkers, uint32_t item_base) {
    PieceMap pm = make_piece_map(row_groups, nseg);
    pm.item_base = item_base;
    const uint32_t grid = set_walkers(pm, walkers);
    std::vector<uint8_t> seen((size_t)walkers * row_groups * nseg, 0);
    for (uint32_t b = 0; b < grid; ++b) {
        uint32_t item, rg, seg;
        const bool live = locate_piece(pm, b, item, rg, seg);
        const uint32_t unit = (b / (8u * nseg)) * 8u + b % 8u, want_seg = (b / 8u) % nseg;
        if (live != (unit < walkers * row_groups)) return printf("live? rg=%u ns=%u w=%u b=%u\n", row_groups, nseg, walkers, b), 1;
        if (!live) continue;
        if (seg != want_seg || rg != unit % row_groups || item != item_base + unit / row_groups)
            return printf("place rg=%u ns=%u w=%u b=%u: item %u rg %u seg %u\n", row_groups, nseg, walkers, b, item, rg, seg), 1;
        uint8_t& s = seen[((size_t)(item - item_base) * row_groups + rg) * nseg + seg];
        if (s) return printf("twice rg=%u ns=%u w=%u b=%u\n", row_groups, nseg, walkers, b), 1;
        s = 1;
        // the segments of a unit are 8 workgroups apart (same XCD)
        if (seg > 0) {
            uint32_t i2, r2, s2;
            if (!locate_piece(pm, b - 8u, i2, r2, s2) || i2 != item || r2 != rg || s2 != seg - 1u)
                return printf("neighbour rg=%u ns=%u w=%u b=%u\n", row_groups, nseg, walkers, b), 1;
        }
    }
    for (uint8_t s : seen) if (!s) return printf("missing rg=%u ns=%u w=%u\n", row_groups, nseg, walkers), 1;
    if (piece_stride(pm) != walkers) return printf("stride rg=%u ns=%u w=%u\n", row_groups, nseg, walkers), 1;
    return 0;
}

// at the largest launch the host allows, the reciprocal divisions still agree with real ones (sampled: the ends and a stride)
static int check_limit(uint32_t row_groups, uint32_t nseg) {
    PieceMap pm = make_piece_map(row_groups, nseg);
    const uint32_t most = most_items(pm);
    if (most == 0) return printf("most = 0 rg=%u ns=%u\n", row_groups, nseg), 1;
    const uint32_t grid = set_walkers(pm, most);
    if ((uint64_t)most * row_groups * nseg > 0xffffffffull) return printf("grid overflow rg=%u ns=%u\n", row_groups, nseg), 1;
    const uint32_t step = grid / 50000u + 1u;
    for (uint64_t bb = 0; bb < grid; bb += (bb < 70000 || bb + 70000 > grid) ? 1 : step) {
        const uint32_t b = (uint32_t)bb;
        uint32_t item, rg, seg;
        const bool live = locate_piece(pm, b, item, rg, seg);
        const uint32_t unit = (b / (8u * nseg)) * 8u + b % 8u;
        if (live != (unit < pm.units)) return printf("limit live rg=%u ns=%u b=%u\n", row_groups, nseg, b), 1;
        if (live && (seg != (b / 8u) % nseg || rg != unit % row_groups || item != unit / row_groups))
            return printf("limit place rg=%u ns=%u b=%u of %u\n", row_groups, nseg, b, grid), 1;
    }
    return 0;
}

int main() {
    unsigned cases = 0;
    for (uint32_t rg = 1; rg <= 40; ++rg)
        for (uint32_t ns = 1; ns <= 14; ++ns)
            for (uint32_t w : {1u, 2u, 3u, 7u, 8u, 9u, 64u, 129u}) {
                if (check_launch(rg, ns, w, (rg * 7u + ns) % 5u)) return 1;
                ++cases;
            }
    for (uint32_t rg : {1u, 2u, 3u, 4u, 5u, 17u, 64u, 256u})
        for (uint32_t ns : {1u, 2u, 3u, 13u, 26u, 103u}) {
            if (check_limit(rg, ns)) return 1;
            ++cases;
        }
    printf("ok %u\n", cases);
    return 0;
}
