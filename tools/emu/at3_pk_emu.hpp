// DEVELOPMENT HARNESS ONLY (tools/emu): host twin of atracdenc_amd/csrc/at3_pk.hpp - the same operations, one at a
// time, in plain C++. The emulator build pre-includes this file (-include), which also claims the product header's
// include guard, so the kernel sources themselves carry no emulator branch.
#ifndef AT3_PK_HPP
#define AT3_PK_HPP
#include <hip/hip_runtime.h>

namespace at3 {

typedef float f2 __attribute__((ext_vector_type(2)));

inline f2 mk2(float x, float y)
{
    f2 v;
    v.x = x;
    v.y = y;
    return v;
}
inline f2 pk_cmul(f2 a, f2 w) { return mk2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x); }
inline f2 pk_add_ib(f2 a, f2 b) { return mk2(a.x - b.y, a.y + b.x); }
inline f2 pk_sub_ib(f2 a, f2 b) { return mk2(a.x + b.y, a.y - b.x); }
inline int opaque_lane_value(int v) { return v; }
inline int count_below8(float4 c0, float4 c1, float key, int r)
{
    return r + (c0.x < key) + (c0.y < key) + (c0.z < key) + (c0.w < key) + (c1.x < key) + (c1.y < key) + (c1.z < key) + (c1.w < key);
}

}  // namespace at3
#define AT3_WAVES_PER_EU(n)
#endif
