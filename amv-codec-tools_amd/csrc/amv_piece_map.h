// amv_piece_map.h -- which piece of which item a reconstruction workgroup works on, from its number alone (device side),
// and the launch that goes with it (host side).  No HIP in here: tests/c/piece_map_test.cc checks the arithmetic on the CPU.
#pragma once
#include <cstdint>
#include <cstdlib>

#if defined(__HIPCC__)
#define AMV_HD __host__ __device__
#else
#define AMV_HD
#endif

namespace amv {

// Where a workgroup works, from its number alone.  The launch is one-dimensional.  A unit is a workgroup's rows of
// one item, (item, row group), numbered item-major; a unit's segments are workgroups b, b + 8, b + 16, ...:
//     b = ((unit / 8) * nseg + segment) * 8 + unit % 8.
// Why: the two segments of a 320-pixel row share a 128-byte line of the picture (byte 480 of 960).  Written a whole
// launch apart -- all frames' first segments, then all second ones: the order of a 3-D grid with the item in x -- that
// line went to memory twice, half-filled (320x240: 5.84 ms per 64 000 frames; side by side in time: 5.66); and blocks b
// and b + 8 share an XCD, hence an L2, so that the halves meet before they leave it (5.57).
// The divisions are by numbers the host knows: it sends their reciprocals (floor(b / d) is the high word of
// b * ceil(2^32 / d) while b * d < 2^32; the launch goes in parts where that would not hold).
struct PieceMap {
    uint32_t row_groups, magic_rg;   // workgroups of an item along its MCU rows
    uint32_t nseg, magic_8ns;        // segments per MCU row; reciprocal of 8 * nseg
    uint32_t units;                  // of this launch
    uint32_t item_base;              // first item of this launch
};
AMV_HD inline uint32_t mul_hi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
AMV_HD inline uint32_t div_by(uint32_t b, uint32_t d, uint32_t magic) { return d == 1u ? b : mul_hi(b, magic); }

// false: a workgroup past the launch's last unit (the launch is rounded up to whole octets of units)
AMV_HD inline bool locate_piece(const PieceMap& pm, uint32_t b, uint32_t& item0, uint32_t& row_group, uint32_t& seg) {
    const uint32_t octet = pm.nseg == 1u ? b >> 3 : mul_hi(b, pm.magic_8ns);   // unit / 8
    const uint32_t within = b - octet * 8u * pm.nseg;
    const uint32_t unit = octet * 8u + (within & 7u);
    seg = within >> 3;
    const uint32_t local = div_by(unit, pm.row_groups, pm.magic_rg);
    row_group = unit - local * pm.row_groups;
    item0 = pm.item_base + local;
    return unit < pm.units;
}
// items between the steps of a workgroup that walks (a round launch)
AMV_HD inline uint32_t piece_stride(const PieceMap& pm) { return pm.units / pm.row_groups; }

// host side: the map of a launch over `walkers` items (set_walkers returns the grid), and how many items one launch may
// take ((workgroup number) * 8 * nseg and (unit number) * row_groups stay below 2^32)
inline PieceMap make_piece_map(uint32_t row_groups, uint32_t nseg) {
    auto magic = [](uint32_t d) { return (uint32_t)((0x100000000ull + d - 1u) / d); };   // (d = 1: unused)
    return PieceMap{row_groups, magic(row_groups), nseg, magic(8u * nseg), 0u, 0u};
}
inline uint32_t set_walkers(PieceMap& pm, uint32_t walkers) {
    pm.units = walkers * pm.row_groups;
    return (pm.units + 7u) / 8u * 8u * pm.nseg;
}
inline uint32_t most_items(const PieceMap& pm) {
    const uint32_t widest = 8u * pm.nseg > pm.row_groups ? 8u * pm.nseg : pm.row_groups;
    const uint32_t most = (0xffffffffu / widest - 8u * pm.nseg) / (pm.row_groups * pm.nseg);
    if (const char* e = getenv("AMVHIP_RECON_MOST")) {   // test knob: launch in parts of this many items
        const long v = atol(e);
        if (v > 0 && (unsigned long)v < most) return (uint32_t)v;
    }
    return most;
}

}  // namespace amv
