// Host check of the reach-mask packing of Geom::rect (splatfields_amd/csrc/common.h: rect_pack / rect_clean / rect_mask16):
// every 12-bit tile rectangle and every 16-bit mask word survive the round trip, and a rectangle stored without a mask reads
// back as "no mask".  Prints: cases, failures (must be 0).
#include <cstdio>
#include <cstdlib>
#include <random>
#include "../../splatfields_amd/csrc/common.h"

int main(int argc, char** argv) {
    const int cases = argc > 1 ? atoi(argv[1]) : 1000000;
    std::mt19937 rng(4242);
    long long bad = 0;
    for (int it = 0; it < cases; ++it) {
        const unsigned x0 = rng() % 4096u, y0 = rng() % 4096u;
        const unsigned x1 = x0 + rng() % (4096u - x0), y1 = y0 + rng() % (4096u - y0);   // <= 4095
        const ushort4 r = make_ushort4((unsigned short)x0, (unsigned short)y0, (unsigned short)x1, (unsigned short)y1);
        const uint32_t m16 = it % 5 == 0 ? 0u : (sr::kRectMasked | (rng() & 0x7fffu));
        const ushort4 p = sr::rect_pack(r, m16);
        const ushort4 c = sr::rect_clean(p);
        if (c.x != r.x || c.y != r.y || c.z != r.z || c.w != r.w || sr::rect_mask16(p) != m16) ++bad;
        if (m16 == 0u && (sr::rect_mask16(p) & sr::kRectMasked)) ++bad;
    }
    // the largest coordinate the C ABI admits
    const ushort4 e = sr::rect_pack(make_ushort4(4095, 4095, 4095, 4095), 0xffffu);
    const ushort4 ec = sr::rect_clean(e);
    if (ec.x != 4095 || ec.w != 4095 || sr::rect_mask16(e) != 0xffffu || sr::kMaxTilesPerSide != 4095) ++bad;
    printf("%d %lld\n", cases, bad);
    return bad ? 1 : 0;
}
