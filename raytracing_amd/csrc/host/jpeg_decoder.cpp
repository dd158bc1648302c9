// JPEG decoder of the host layer: baseline and progressive Huffman, 8-bit, 1 or 3 components.
//
// The renderer's texels are whatever the reference's decoder (stb_image v2.19, vendored at
// include/stb_image.h in the reference) produces: two decoders that are both "correct" still differ
// by a level or two per sample in the inverse DCT, the chroma upsampling filter and the colour
// conversion, and every such difference would show in a frame.  This file is an independent
// implementation of the same pipeline with the same integer arithmetic, stage by stage:
//   - coefficients: Huffman decoding per ITU T.81 (F.2.2, G.1.2 for progressive scans); the
//     dequantised value is truncated to 16 bits (stb_image.h:1969,1999,2810-2815);
//   - inverse DCT: the 12-bit fixed-point "islow" butterfly with a column pass that keeps two extra
//     bits and a row pass that rounds at bit 17 and re-centres by 128 (stb_image.h:2163-2261);
//   - upsampling: for 2x horizontal and/or vertical subsampling the 3:1 "triangle" filters with
//     +2 >> 2 and +8 >> 4 rounding, nearest neighbour for every other ratio (stb_image.h:3164-3363),
//     walked with the same near/far row schedule (stb_image.h:3603-3645);
//   - colour: YCbCr -> RGB in 20-bit fixed point with the constants rounded to 12 bits and the
//     Cb term of green masked to its upper 16 bits (stb_image.h:3365-3391).
// tests/test_jpeg_decoder.py compares it with the reference's decoder on every JPEG the reference
// ships (when oracle/_ref has been built) and with committed SHA-256 digests of those decodes.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/rt_host.h"
#include "host_error.hpp"

namespace {

struct JpegError {
    const char *what;
};
[[noreturn]] void bad(const char *what) { throw JpegError{what}; }

// zig-zag position -> row-major position
const uint8_t kDezigzag[64 + 15] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33,
                                    40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36,
                                    29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54,
                                    47, 55, 62, 63,
                                    // run lengths in damaged data may step past 63
                                    63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

struct Huffman {
    // canonical code tables of T.81 Annex C / F.2.2.3
    uint8_t size[257];
    uint8_t values[256];
    int32_t mincode[18], maxcode[18], valptr[18];
    int16_t lookup[512];  // 9-bit prefix -> (size << 8 | symbol), -1 if longer
    bool defined = false;

    void build(const int counts[16], const uint8_t *vals, int nvals) {
        int k = 0;
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < counts[i]; ++j) size[k++] = (uint8_t)(i + 1);
        size[k] = 0;
        const int total = k;
        if (total > 256 || total != nvals) bad("bad code lengths");
        memcpy(values, vals, (size_t)nvals);
        int code = 0;
        k = 0;
        uint16_t codes[256];
        for (int len = 1; len <= 16; ++len) {
            valptr[len] = k;
            mincode[len] = code;
            while (k < total && size[k] == len) codes[k++] = (uint16_t)code++;
            if (code - 1 >= (1 << len) && k > valptr[len]) bad("bad code lengths");
            maxcode[len] = k > valptr[len] ? code - 1 : -1;
            code <<= 1;
        }
        for (int i = 0; i < 512; ++i) lookup[i] = -1;
        for (int i = 0; i < total; ++i) {
            const int s = size[i];
            if (s <= 9) {
                const int first = codes[i] << (9 - s), n = 1 << (9 - s);
                for (int j = 0; j < n; ++j) lookup[first + j] = (int16_t)((s << 8) | values[i]);
            }
        }
        defined = true;
    }
};

// Entropy-coded segment reader: MSB-first bits, 0xFF00 unstuffing.  Once a marker is met inside the
// segment it is remembered and the stream continues as zeros, which is what the reference's reader
// feeds its decoder (stb_image.h:1823-1839).
struct BitReader {
    const uint8_t *p, *end;
    uint32_t buf = 0;
    int bits = 0;
    int marker = -1;  // marker byte seen while filling, -1 if none
    bool nomore = false;

    BitReader(const uint8_t *b, const uint8_t *e) : p(b), end(e) {}
    int byte() { return p < end ? *p++ : 0; }
    void reset() {
        buf = 0, bits = 0, marker = -1, nomore = false;
    }
    void fill() {
        do {
            unsigned b = nomore ? 0u : (unsigned)byte();
            if (b == 0xff) {
                int c = byte();
                while (c == 0xff) c = byte();
                if (c != 0) {
                    marker = c;
                    nomore = true;
                    return;
                }
            }
            buf |= b << (24 - bits);
            bits += 8;
        } while (bits <= 24);
    }
    int getBits(int n) {  // n in 0..16
        if (n == 0) return 0;
        if (bits < n) fill();
        const int v = (int)(buf >> (32 - n));
        buf <<= n;
        bits -= n;
        return v;
    }
    int getBit() {
        if (bits < 1) fill();
        const int v = (int)(buf >> 31);
        buf <<= 1;
        --bits;
        return v;
    }
    int decode(const Huffman &h) {
        if (bits < 16) fill();
        const int pre = lookup9();
        const int16_t e = h.lookup[pre];
        if (e >= 0) {
            const int s = e >> 8;
            if (s > bits) bad("bad huffman code");
            buf <<= s;
            bits -= s;
            return e & 255;
        }
        int code = (int)(buf >> (32 - 9));
        for (int len = 10; len <= 16; ++len) {
            code = (int)(buf >> (32 - len));
            if (h.maxcode[len] >= 0 && code <= h.maxcode[len] && code >= h.mincode[len]) {
                if (len > bits) bad("bad huffman code");
                buf <<= len;
                bits -= len;
                return h.values[h.valptr[len] + code - h.mincode[len]];
            }
        }
        bad("bad huffman code");
    }
    int lookup9() const { return (int)(buf >> (32 - 9)); }
    // T.81 F.2.2.1 RECEIVE + EXTEND
    int receiveExtend(int n) {
        if (n == 0) return 0;
        const int v = getBits(n);
        return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, hd = 0, ha = 0, dc_pred = 0;
    int x = 0, y = 0, w2 = 0, h2 = 0;  // effective size in samples; padded plane size
    std::vector<uint8_t> data;          // w2 x h2 samples
    std::vector<int16_t> coeff;         // progressive: (w2/8) x (h2/8) blocks of 64
    int coeff_w = 0;
};

inline uint8_t clampByte(int x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }

// 12-bit fixed-point constants of the inverse DCT, rounded as int(x * 4096 + 0.5)
constexpr int fx(double x) { return (int)(x * 4096 + 0.5); }

// One 8-point inverse DCT pass (the even/odd split of the "islow" algorithm, constants scaled by
// 4096).  Returns the even part in e[0..3] and the odd part in o[0..3]; output k is e[k] + o[3-k]
// for k < 4 and e[7-k] - o[k-4] for k >= 4, before the pass-specific rounding and shift.
// (The arithmetic is 32-bit two's complement that wraps, written out in unsigned: a damaged file can hold coefficients
// that overflow it, and the values of an intact one are the same either way.)
inline void idct8(int i0, int i1, int i2, int i3, int i4, int i5, int i6, int i7, int e[4], int o[4]) {
    using U = uint32_t;
    const U s0 = (U)i0, s1 = (U)i1, s2 = (U)i2, s3 = (U)i3, s4 = (U)i4, s5 = (U)i5, s6 = (U)i6, s7 = (U)i7;
    constexpr auto k = [](double x) { return (U)fx(x); };
    // even part: rotate (s2, s6), butterfly (s0, s4)
    const U z = (s2 + s6) * k(0.5411961f);
    const U r2 = z + s6 * k(-1.847759065f);
    const U r3 = z + s2 * k(0.765366865f);
    const U a0 = (s0 + s4) * 4096u, a1 = (s0 - s4) * 4096u;
    e[0] = (int)(a0 + r3), e[3] = (int)(a0 - r3), e[1] = (int)(a1 + r2), e[2] = (int)(a1 - r2);
    // odd part on (s7, s5, s3, s1)
    const U q3 = s7 + s3, q4 = s5 + s1, q1 = s7 + s1, q2 = s5 + s3;
    const U z5 = (q3 + q4) * k(1.175875602f);
    const U m1 = z5 + q1 * k(-0.899976223f), m2 = z5 + q2 * k(-2.562915447f);
    const U m3 = q3 * k(-1.961570560f), m4 = q4 * k(-0.390180644f);
    o[3] = (int)(s1 * k(1.501321110f) + (m1 + m4));
    o[2] = (int)(s3 * k(3.072711026f) + (m2 + m3));
    o[1] = (int)(s5 * k(2.053119869f) + (m2 + m4));
    o[0] = (int)(s7 * k(0.298631336f) + (m1 + m3));
}

// a + b and a - b of 32-bit two's complement values that wrap
inline int wrapAdd(int a, int b) { return (int)((uint32_t)a + (uint32_t)b); }
inline int wrapSub(int a, int b) { return (int)((uint32_t)a - (uint32_t)b); }

void idctBlock(uint8_t *out, int stride, const int16_t d[64]) {
    int val[64];
    int e[4], o[4];
    for (int i = 0; i < 8; ++i) {  // columns
        const int16_t *c = d + i;
        int *v = val + i;
        if (c[8] == 0 && c[16] == 0 && c[24] == 0 && c[32] == 0 && c[40] == 0 && c[48] == 0 && c[56] == 0) {
            // a column with only a DC term is that term scaled to the pass's 2 extra bits
            const int dc = c[0] * 4;  // (16 bits times 4: no overflow)
            for (int k = 0; k < 8; ++k) v[8 * k] = dc;
            continue;
        }
        idct8(c[0], c[8], c[16], c[24], c[32], c[40], c[48], c[56], e, o);
        // drop the 12 constant bits but keep 2 bits of extra precision: round at bit 9
        for (int k = 0; k < 4; ++k) {
            const int x = wrapAdd(e[k], 512);
            v[8 * k] = wrapAdd(x, o[3 - k]) >> 10;
            v[8 * (7 - k)] = wrapSub(x, o[3 - k]) >> 10;
        }
    }
    for (int i = 0; i < 8; ++i) {  // rows
        const int *v = val + 8 * i;
        uint8_t *row = out + (size_t)stride * i;
        idct8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], e, o);
        // 12 constant bits + 2 kept bits + 3 bits from the two sqrt(8) scalings = 17: round there,
        // and move -128..127 to 0..255 in the same addition
        for (int k = 0; k < 4; ++k) {
            const int x = wrapAdd(e[k], 65536 + (128 << 17));
            row[k] = clampByte(wrapAdd(x, o[3 - k]) >> 17);
            row[7 - k] = clampByte(wrapSub(x, o[3 - k]) >> 17);
        }
    }
}

struct Decoder {
    const uint8_t *base, *p, *end;
    int width = 0, height = 0, ncomp = 0;
    bool progressive = false, jfif = false;
    int app14_transform = -1, rgb_ids = 0;
    Component comp[4];
    uint16_t dequant[4][64];
    Huffman hdc[4], hac[4];
    int h_max = 1, v_max = 1, mcu_x = 0, mcu_y = 0;
    int restart_interval = 0;
    // scan state
    int scan_n = 0, order[4];
    int spec_start = 0, spec_end = 63, succ_high = 0, succ_low = 0, eob_run = 0, todo = 0;
    int pending_marker = -1;  // marker found by the entropy reader

    Decoder(const uint8_t *b, size_t n) : base(b), p(b), end(b + n) { memset(dequant, 0, sizeof dequant); }

    int get8() { return p < end ? *p++ : 0; }
    int get16() {
        const int hi = get8();
        return (hi << 8) | get8();
    }
    bool eof() const { return p >= end; }
    void skip(int n) {
        if (n < 0 || n > end - p)
            p = end;
        else
            p += n;
    }

    // next marker: 0xFF, any number of 0xFF fill bytes, then the code; -1 when the next byte is not 0xFF
    int getMarker() {
        if (pending_marker >= 0) {
            const int m = pending_marker;
            pending_marker = -1;
            return m;
        }
        int x = get8();
        if (x != 0xff) return -1;
        while (x == 0xff) x = get8();
        return x;
    }

    void processMarker(int m) {
        switch (m) {
            case -1: bad("expected marker");
            case 0xDD:  // restart interval
                if (get16() != 4) bad("bad DRI len");
                restart_interval = get16();
                return;
            case 0xDB: {  // quantisation tables
                int L = get16() - 2;
                while (L > 0) {
                    const int q = get8(), prec = q >> 4, t = q & 15;
                    if (prec != 0 && prec != 1) bad("bad DQT type");
                    if (t > 3) bad("bad DQT table");
                    for (int i = 0; i < 64; ++i) dequant[t][kDezigzag[i]] = (uint16_t)(prec ? get16() : get8());
                    L -= prec ? 129 : 65;
                }
                if (L != 0) bad("bad DQT len");
                return;
            }
            case 0xC4: {  // Huffman tables
                int L = get16() - 2;
                while (L > 0) {
                    const int q = get8(), tc = q >> 4, th = q & 15;
                    if (tc > 1 || th > 3) bad("bad DHT header");
                    int counts[16], n = 0;
                    for (int i = 0; i < 16; ++i) n += counts[i] = get8();
                    if (n > 256) bad("bad DHT header");
                    uint8_t vals[256];
                    for (int i = 0; i < n; ++i) vals[i] = (uint8_t)get8();
                    (tc == 0 ? hdc[th] : hac[th]).build(counts, vals, n);
                    L -= 17 + n;
                }
                if (L != 0) bad("bad DHT len");
                return;
            }
        }
        if ((m >= 0xE0 && m <= 0xEF) || m == 0xFE) {  // APPn / COM
            int L = get16();
            if (L < 2) bad(m == 0xFE ? "bad COM len" : "bad APP len");
            L -= 2;
            if (m == 0xE0 && L >= 5) {
                static const uint8_t tag[5] = {'J', 'F', 'I', 'F', 0};
                bool ok = true;
                for (int i = 0; i < 5; ++i) ok &= get8() == tag[i];
                L -= 5;
                if (ok) jfif = true;
            } else if (m == 0xEE && L >= 12) {
                static const uint8_t tag[6] = {'A', 'd', 'o', 'b', 'e', 0};
                bool ok = true;
                for (int i = 0; i < 6; ++i) ok &= get8() == tag[i];
                L -= 6;
                if (ok) {
                    get8(), get16(), get16();
                    app14_transform = get8();
                    L -= 6;
                }
            }
            skip(L);
            return;
        }
        bad("unknown marker");
    }

    void frameHeader() {
        const int Lf = get16();
        if (Lf < 11) bad("bad SOF len");
        if (get8() != 8) bad("only 8-bit");
        height = get16();
        if (height == 0) bad("no header height");
        width = get16();
        if (width == 0) bad("0 width");
        ncomp = get8();
        if (ncomp != 3 && ncomp != 1 && ncomp != 4) bad("bad component count");
        if (Lf != 8 + 3 * ncomp) bad("bad SOF len");
        rgb_ids = 0;
        for (int i = 0; i < ncomp; ++i) {
            static const uint8_t rgb[3] = {'R', 'G', 'B'};
            comp[i].id = get8();
            if (ncomp == 3 && comp[i].id == rgb[i]) ++rgb_ids;
            const int q = get8();
            comp[i].h = q >> 4, comp[i].v = q & 15;
            if (!comp[i].h || comp[i].h > 4) bad("bad H");
            if (!comp[i].v || comp[i].v > 4) bad("bad V");
            comp[i].tq = get8();
            if (comp[i].tq > 3) bad("bad TQ");
        }
        if ((uint64_t)width * height * ncomp > (1ull << 31) - 1) bad("too large");
        h_max = v_max = 1;
        for (int i = 0; i < ncomp; ++i) {
            if (comp[i].h > h_max) h_max = comp[i].h;
            if (comp[i].v > v_max) v_max = comp[i].v;
        }
        mcu_x = (width + h_max * 8 - 1) / (h_max * 8);
        mcu_y = (height + v_max * 8 - 1) / (v_max * 8);
        for (int i = 0; i < ncomp; ++i) {
            Component &c = comp[i];
            c.x = (width * c.h + h_max - 1) / h_max;
            c.y = (height * c.v + v_max - 1) / v_max;
            c.w2 = mcu_x * c.h * 8;
            c.h2 = mcu_y * c.v * 8;
            c.data.assign((size_t)c.w2 * c.h2, 0);
            if (progressive) {
                c.coeff_w = c.w2 / 8;
                c.coeff.assign((size_t)c.w2 * c.h2, 0);
            }
        }
    }

    void scanHeader() {
        const int Ls = get16();
        scan_n = get8();
        if (scan_n < 1 || scan_n > 4 || scan_n > ncomp) bad("bad SOS component count");
        if (Ls != 6 + 2 * scan_n) bad("bad SOS len");
        for (int i = 0; i < scan_n; ++i) {
            const int id = get8(), q = get8();
            int which = 0;
            while (which < ncomp && comp[which].id != id) ++which;
            if (which == ncomp) bad("bad SOS component");
            comp[which].hd = q >> 4;
            comp[which].ha = q & 15;
            if (comp[which].hd > 3) bad("bad DC huff");
            if (comp[which].ha > 3) bad("bad AC huff");
            order[i] = which;
        }
        spec_start = get8();
        spec_end = get8();
        const int aa = get8();
        succ_high = aa >> 4, succ_low = aa & 15;
        if (progressive) {
            if (spec_start > 63 || spec_end > 63 || spec_start > spec_end || succ_high > 13 || succ_low > 13) bad("bad SOS");
        } else {
            if (spec_start != 0 || succ_high != 0 || succ_low != 0) bad("bad SOS");
            spec_end = 63;
        }
    }

    void resetEntropy(BitReader &br) {
        br.reset();
        for (Component &c : comp) c.dc_pred = 0;
        todo = restart_interval ? restart_interval : 0x7fffffff;
        eob_run = 0;
    }

    // sequential block: DC difference, then (run, size) AC pairs; every value is multiplied by its
    // quantiser and truncated to 16 bits on the spot
    void decodeBlock(BitReader &br, int16_t data[64], Component &c) {
        const Huffman &dc = hdc[c.hd], &ac = hac[c.ha];
        if (!dc.defined || !ac.defined) bad("missing huffman table");
        const uint16_t *dq = dequant[c.tq];
        const int t = br.decode(dc);
        memset(data, 0, 64 * sizeof(int16_t));
        if (t > 16) bad("bad huffman code");
        const int diff = t ? br.receiveExtend(t) : 0;
        c.dc_pred = wrapAdd(c.dc_pred, diff);
        data[0] = (int16_t)((uint32_t)c.dc_pred * dq[0]);
        int k = 1;
        do {
            const int rs = br.decode(ac);
            const int s = rs & 15, r = rs >> 4;
            if (s == 0) {
                if (rs != 0xf0) break;
                k += 16;
            } else {
                k += r;
                const int zig = kDezigzag[k++];
                data[zig] = (int16_t)(br.receiveExtend(s) * dq[zig]);
            }
        } while (k < 64);
    }

    void progDc(BitReader &br, int16_t data[64], Component &c) {
        if (spec_end != 0) bad("can't merge dc and ac");
        if (succ_high == 0) {
            const Huffman &dc = hdc[c.hd];
            if (!dc.defined) bad("missing huffman table");
            memset(data, 0, 64 * sizeof(int16_t));
            const int t = br.decode(dc);
            if (t > 16) bad("bad huffman code");
            const int diff = t ? br.receiveExtend(t) : 0;
            c.dc_pred = wrapAdd(c.dc_pred, diff);
            data[0] = (int16_t)((uint32_t)c.dc_pred << succ_low);
        } else if (br.getBit()) {
            data[0] += (int16_t)(1 << succ_low);
        }
    }

    void progAc(BitReader &br, int16_t data[64], Component &c) {
        if (spec_start == 0) bad("can't merge dc and ac");
        const Huffman &ac = hac[c.ha];
        if (!ac.defined) bad("missing huffman table");
        if (succ_high == 0) {  // first pass over this band (T.81 G.1.2.2)
            const int shift = succ_low;
            if (eob_run) {
                --eob_run;
                return;
            }
            int k = spec_start;
            do {
                const int rs = br.decode(ac);
                const int s = rs & 15, r = rs >> 4;
                if (s == 0) {
                    if (r < 15) {
                        eob_run = 1 << r;
                        if (r) eob_run += br.getBits(r);
                        --eob_run;
                        break;
                    }
                    k += 16;
                } else {
                    k += r;
                    const int zig = kDezigzag[k++];
                    data[zig] = (int16_t)((uint32_t)br.receiveExtend(s) << shift);
                }
            } while (k <= spec_end);
        } else {  // refinement pass (G.1.2.3)
            const int16_t bit = (int16_t)(1 << succ_low);
            auto refine = [&](int16_t *q) {
                if (br.getBit() && (*q & bit) == 0) {
                    if (*q > 0)
                        *q += bit;
                    else
                        *q -= bit;
                }
            };
            if (eob_run) {
                --eob_run;
                for (int k = spec_start; k <= spec_end; ++k) {
                    int16_t *q = &data[kDezigzag[k]];
                    if (*q != 0) refine(q);
                }
            } else {
                int k = spec_start;
                do {
                    const int rs = br.decode(ac);
                    int s = rs & 15, r = rs >> 4;
                    if (s == 0) {
                        if (r < 15) {
                            eob_run = (1 << r) - 1;
                            if (r) eob_run += br.getBits(r);
                            r = 64;  // run to the end of the band
                        }
                    } else {
                        if (s != 1) bad("bad huffman code");
                        s = br.getBit() ? bit : -bit;
                    }
                    while (k <= spec_end) {
                        int16_t *q = &data[kDezigzag[k++]];
                        if (*q != 0) {
                            refine(q);
                        } else {
                            if (r == 0) {
                                *q = (int16_t)s;
                                break;
                            }
                            --r;
                        }
                    }
                } while (k <= spec_end);
            }
        }
    }

    // true: carry on; false: the interval ended without a restart marker, give up on this scan
    bool countRestart(BitReader &br) {
        if (--todo <= 0) {
            if (br.bits < 24) br.fill();
            if (!(br.marker >= 0xd0 && br.marker <= 0xd7)) return false;
            resetEntropy(br);
        }
        return true;
    }

    void entropyCodedData() {
        BitReader br(p, end);
        resetEntropy(br);
        int16_t block[64];
        auto run = [&]() {
            if (scan_n == 1) {
                // one component: blocks in raster order over its own size, not the MCU-padded one
                Component &c = comp[order[0]];
                const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
                for (int j = 0; j < h; ++j)
                    for (int i = 0; i < w; ++i) {
                        if (!progressive) {
                            decodeBlock(br, block, c);
                            idctBlock(c.data.data() + (size_t)c.w2 * j * 8 + i * 8, c.w2, block);
                        } else {
                            int16_t *data = c.coeff.data() + 64 * ((size_t)i + (size_t)j * c.coeff_w);
                            if (spec_start == 0)
                                progDc(br, data, c);
                            else
                                progAc(br, data, c);
                        }
                        if (!countRestart(br)) return;
                    }
            } else {
                for (int j = 0; j < mcu_y; ++j)
                    for (int i = 0; i < mcu_x; ++i) {
                        for (int k = 0; k < scan_n; ++k) {
                            Component &c = comp[order[k]];
                            for (int y = 0; y < c.v; ++y)
                                for (int x = 0; x < c.h; ++x) {
                                    const int bx = i * c.h + x, by = j * c.v + y;
                                    if (!progressive) {
                                        decodeBlock(br, block, c);
                                        idctBlock(c.data.data() + (size_t)c.w2 * by * 8 + bx * 8, c.w2, block);
                                    } else {
                                        progDc(br, c.coeff.data() + 64 * ((size_t)bx + (size_t)by * c.coeff_w), c);
                                    }
                                }
                        }
                        if (!countRestart(br)) return;
                    }
            }
        };
        run();
        p = br.p;
        pending_marker = br.marker;
        if (pending_marker < 0) {
            // stray bytes after the entropy data: move to the next 0xFF and take what follows it
            while (!eof()) {
                if (get8() == 255) {
                    pending_marker = get8();
                    break;
                }
            }
        }
    }

    void finishProgressive() {
        for (int n = 0; n < ncomp; ++n) {
            Component &c = comp[n];
            const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
            const uint16_t *dq = dequant[c.tq];
            for (int j = 0; j < h; ++j)
                for (int i = 0; i < w; ++i) {
                    int16_t *data = c.coeff.data() + 64 * ((size_t)i + (size_t)j * c.coeff_w);
                    for (int k = 0; k < 64; ++k) data[k] = (int16_t)(data[k] * dq[k]);
                    idctBlock(c.data.data() + (size_t)c.w2 * j * 8 + i * 8, c.w2, data);
                }
        }
    }

    void decodePlanes() {
        int m = getMarker();
        if (m != 0xd8) bad("no SOI");
        m = getMarker();
        while (!(m == 0xc0 || m == 0xc1 || m == 0xc2)) {
            processMarker(m);
            m = getMarker();
            while (m == -1) {
                if (eof()) bad("no SOF");
                m = getMarker();
            }
        }
        progressive = m == 0xc2;
        frameHeader();
        m = getMarker();
        while (m != 0xd9) {
            if (m == 0xda) {
                scanHeader();
                entropyCodedData();
            } else if (m == 0xdc) {
                const int Ld = get16(), NL = get16();
                if (Ld != 4) bad("bad DNL len");
                if (NL != height) bad("bad DNL height");
            } else {
                processMarker(m);
            }
            m = getMarker();
        }
        if (progressive) finishProgressive();
    }
};

// ---- chroma upsampling: one output row from a near and a far input row ----
using ResampleFn = const uint8_t *(*)(uint8_t *out, const uint8_t *near_, const uint8_t *far_, int w, int hs);

const uint8_t *resample1(uint8_t *, const uint8_t *near_, const uint8_t *, int, int) { return near_; }
const uint8_t *resampleV2(uint8_t *out, const uint8_t *near_, const uint8_t *far_, int w, int) {
    for (int i = 0; i < w; ++i) out[i] = (uint8_t)((3 * near_[i] + far_[i] + 2) >> 2);
    return out;
}
const uint8_t *resampleH2(uint8_t *out, const uint8_t *in, const uint8_t *, int w, int) {
    if (w == 1) {
        out[0] = out[1] = in[0];
        return out;
    }
    out[0] = in[0];
    out[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
    int i;
    for (i = 1; i < w - 1; ++i) {
        const int n = 3 * in[i] + 2;
        out[i * 2 + 0] = (uint8_t)((n + in[i - 1]) >> 2);
        out[i * 2 + 1] = (uint8_t)((n + in[i + 1]) >> 2);
    }
    out[i * 2 + 0] = (uint8_t)((in[w - 2] * 3 + in[w - 1] + 2) >> 2);
    out[i * 2 + 1] = in[w - 1];
    return out;
}
const uint8_t *resampleHV2(uint8_t *out, const uint8_t *near_, const uint8_t *far_, int w, int) {
    if (w == 1) {
        out[0] = out[1] = (uint8_t)((3 * near_[0] + far_[0] + 2) >> 2);
        return out;
    }
    int t1 = 3 * near_[0] + far_[0];
    out[0] = (uint8_t)((t1 + 2) >> 2);
    for (int i = 1; i < w; ++i) {
        const int t0 = t1;
        t1 = 3 * near_[i] + far_[i];
        out[i * 2 - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4);
        out[i * 2] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
    }
    out[w * 2 - 1] = (uint8_t)((t1 + 2) >> 2);
    return out;
}
const uint8_t *resampleNearest(uint8_t *out, const uint8_t *in, const uint8_t *, int w, int hs) {
    for (int i = 0; i < w; ++i)
        for (int j = 0; j < hs; ++j) out[i * hs + j] = in[i];
    return out;
}

// YCbCr -> RGB: constants rounded to 12 bits then scaled to 20, y rounded at bit 19, the Cb
// contribution to green truncated to its upper 16 bits before the sum
constexpr int fixed20(float x) { return ((int)(x * 4096.0f + 0.5f)) << 8; }
void ycbcrRow(uint8_t *out, const uint8_t *y, const uint8_t *pcb, const uint8_t *pcr, int count) {
    for (int i = 0; i < count; ++i) {
        const int y_fixed = (y[i] << 20) + (1 << 19);
        const int cr = pcr[i] - 128, cb = pcb[i] - 128;
        int r = y_fixed + cr * fixed20(1.40200f);
        int g = y_fixed + (cr * -fixed20(0.71414f)) + (int)((unsigned)(cb * -fixed20(0.34414f)) & 0xffff0000u);
        int b = y_fixed + cb * fixed20(1.77200f);
        r >>= 20, g >>= 20, b >>= 20;
        out[0] = clampByte(r), out[1] = clampByte(g), out[2] = clampByte(b);
        out += 3;
    }
}

uint8_t *decodeJpeg(const uint8_t *bytes, size_t n, int *w, int *h, int *ncomp) {
    Decoder d(bytes, n);
    d.decodePlanes();
    if (d.ncomp == 4) bad("4-component JPEG not supported");
    const int out_n = d.ncomp >= 3 ? 3 : 1;
    const bool is_rgb = d.ncomp == 3 && (d.rgb_ids == 3 || (d.app14_transform == 0 && !d.jfif));
    struct Resampler {
        ResampleFn fn;
        const uint8_t *line0, *line1;
        int hs, vs, w_lores, ystep, ypos;
        std::vector<uint8_t> linebuf;
    } rs[3];
    for (int k = 0; k < d.ncomp; ++k) {
        Resampler &r = rs[k];
        r.linebuf.assign((size_t)d.width + 3 + 8, 0);
        r.hs = d.h_max / d.comp[k].h;
        r.vs = d.v_max / d.comp[k].v;
        r.ystep = r.vs >> 1;
        r.w_lores = (d.width + r.hs - 1) / r.hs;
        r.ypos = 0;
        r.line0 = r.line1 = d.comp[k].data.data();
        if (r.hs == 1 && r.vs == 1) r.fn = resample1;
        else if (r.hs == 1 && r.vs == 2) r.fn = resampleV2;
        else if (r.hs == 2 && r.vs == 1) r.fn = resampleH2;
        else if (r.hs == 2 && r.vs == 2) r.fn = resampleHV2;
        else r.fn = resampleNearest;
    }
    uint8_t *output = (uint8_t *)malloc((size_t)out_n * d.width * d.height);
    if (!output) bad("out of memory");
    for (int j = 0; j < d.height; ++j) {
        uint8_t *out = output + (size_t)out_n * d.width * j;
        const uint8_t *co[3] = {nullptr, nullptr, nullptr};
        for (int k = 0; k < d.ncomp; ++k) {
            Resampler &r = rs[k];
            // the row nearer to this output row gets weight 3; which of the two that is alternates
            const bool y_bot = r.ystep >= (r.vs >> 1);
            co[k] = r.fn(r.linebuf.data(), y_bot ? r.line1 : r.line0, y_bot ? r.line0 : r.line1, r.w_lores, r.hs);
            if (++r.ystep >= r.vs) {
                r.ystep = 0;
                r.line0 = r.line1;
                if (++r.ypos < d.comp[k].y) r.line1 += d.comp[k].w2;
            }
        }
        if (out_n == 3) {
            if (is_rgb) {
                for (int i = 0; i < d.width; ++i) out[3 * i] = co[0][i], out[3 * i + 1] = co[1][i], out[3 * i + 2] = co[2][i];
            } else {
                ycbcrRow(out, co[0], co[1], co[2], d.width);
            }
        } else {
            memcpy(out, co[0], (size_t)d.width);
        }
    }
    *w = d.width, *h = d.height, *ncomp = out_n;
    return output;
}

}  // namespace

extern "C" int rt_decode_jpeg(const char *path, unsigned char **rgb, int *w, int *h, int *ncomp) {
    if (!path || !rgb || !w || !h || !ncomp) return -1;
    *rgb = nullptr, *w = 0, *h = 0, *ncomp = 0;
    FILE *fp = fopen(path, "rb");
    if (!fp) {
        rt::setHostError("can't fopen");  // the reference decoder's failure reason for a missing file
        return -1;
    }
    std::vector<uint8_t> bytes;
    uint8_t buf[1 << 16];
    size_t got;
    while ((got = fread(buf, 1, sizeof buf, fp)) > 0) bytes.insert(bytes.end(), buf, buf + got);
    fclose(fp);
    try {
        *rgb = decodeJpeg(bytes.data(), bytes.size(), w, h, ncomp);
    } catch (const JpegError &e) {
        rt::setHostError(e.what);
        return -1;
    } catch (const std::bad_alloc &) {
        rt::setHostError("outofmem");
        return -1;
    }
    return 0;
}
