/*
 * pt_child_table.h -- look-up tables of the octree walk's node step, made at compile time and shared by the host
 * (table export for tests) and the device (copied into LDS by every workgroup).
 *
 * The children of a node that proc_subtree's loop (src/device_geometry_functions.cuh:159-188) steps through, in the
 * mirrored child numbering, are a function of comparisons between the node's interval parameters alone.  Write
 * M_i = (tM_i, i) and T_i = (t1_i, i) for the mid-plane and exit-plane parameters of axis i (x = 0, y = 1, z = 2) and
 * order such events by value, ties going to the LARGER axis: (v, i) before (w, j) iff v < w or (v == w and i > j).
 * next_node (dgf:116-126: x needs ex < ey and ex < ez; y needs ey <= ex and ey < ez; z otherwise) returns exactly the
 * first of the three exit events of the current child in that order.  The loop starts at the child whose bits are
 * S0 = {i : tM_i < te}, te = max(t0) (find_first_node, dgf:95-110), then keeps adding the axis of the first exit
 * event while that event is a mid plane, and leaves the node when it is an exit plane.  Hence
 *   - axis w outside S0 gets added iff M_w comes before T_j for BOTH other axes j (were some T_j first, the loop would
 *     have left the node before reaching M_w; T_w itself cannot come before M_w since tM_w <= t1_w);
 *   - the axes that get added are added in the order of their M events.
 * In comparisons (the ten bits of the table key, most significant first, the order the walk shifts them in):
 *   9,8,7  sx, sy, sz   tM_i < te
 *   6      ax           tMx < min(t1y, t1z)                 x can be added
 *   5      nxy          t1x < tMy          \  y can be added iff !nxy and pyz  (tMy <= t1x: y wins the tie against x;
 *   4      pyz          tMy < t1z          /                                    tMy < t1z: z wins the tie against y)
 *   3      nz           min(t1x, t1y) < tMz                 z can be added iff !nz
 *   2,1,0  oxy, oxz, oyz   tMx < tMy, tMx < tMz, tMy < tMz  order of the mid-plane events
 * set[key] has bit c set for every child c (mirrored numbering) the loop steps through.  Keys that no real interval
 * produces (a cyclic order) hold an arbitrary value.
 *
 * mir[a << 8 | m]: the child mask m of a node (bit k = child slot k exists) in the mirrored numbering of a ray with
 * mirror bits a (dgf:165: child c of the loop is slot c ^ a): bit c = bit (c ^ a) of m.
 */
#ifndef PT_CHILD_TABLE_H
#define PT_CHILD_TABLE_H

#include <stdint.h>

namespace ptamd {

static constexpr int kChildSetEntries = 1024;
static constexpr int kChildMirEntries = 8 * 256;

struct ChildTables {
    uint8_t set[kChildSetEntries];
    uint8_t mir[kChildMirEntries];
};

constexpr uint8_t child_set_entry(unsigned key) {
    const bool sx = (key >> 9) & 1u, sy = (key >> 8) & 1u, sz = (key >> 7) & 1u;
    const bool ax = (key >> 6) & 1u, nxy = (key >> 5) & 1u, pyz = (key >> 4) & 1u, nz = (key >> 3) & 1u;
    const bool oxy = (key >> 2) & 1u, oxz = (key >> 1) & 1u, oyz = key & 1u;
    const bool can[3] = {ax, !nxy && pyz, !nz};
    /* how many mid-plane events come before each axis' own */
    const int rank[3] = {(oxy ? 0 : 1) + (oxz ? 0 : 1), (oxy ? 1 : 0) + (oyz ? 0 : 1), (oxz ? 1 : 0) + (oyz ? 1 : 0)};
    const unsigned bit[3] = {4u, 2u, 1u};
    unsigned c = (sx ? 4u : 0u) | (sy ? 2u : 0u) | (sz ? 1u : 0u);
    unsigned set = 1u << c;
    for (int r = 0; r < 3; r++)
        for (int i = 0; i < 3; i++)
            if (rank[i] == r && can[i] && !(c & bit[i])) {
                c |= bit[i];
                set |= 1u << c;
            }
    return (uint8_t)set;
}

constexpr uint8_t child_mir_entry(unsigned a, unsigned m) {
    unsigned r = 0;
    for (unsigned c = 0; c < 8; c++) r |= ((m >> (c ^ a)) & 1u) << c;
    return (uint8_t)r;
}

constexpr ChildTables make_child_tables() {
    ChildTables t{};
    for (unsigned k = 0; k < (unsigned)kChildSetEntries; k++) t.set[k] = child_set_entry(k);
    for (unsigned a = 0; a < 8; a++)
        for (unsigned m = 0; m < 256; m++) t.mir[a << 8 | m] = child_mir_entry(a, m);
    return t;
}

} // namespace ptamd
#endif
