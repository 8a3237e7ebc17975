// Colour frames of a sequence, from the .jpg file to the cropped, downsampled image in HBM -- SURVEY.md 8 (f4), reference
// utils.py:441-457 (get_pair_color_imgs: cv2.imread -> cv2.resize(fx = fy = 1/d) -> crop -> BGR2RGB) and dataset.py:148,446-451
// (albumentations Normalize(mean 0.5, std 0.5) + img_to_tensor).
//
// Split the way the format dictates:
//   host    baseline-JPEG parsing and Huffman decoding (inherently serial bit stream) into coefficient blocks, written straight
//           into the caller's staging buffer (pinned memory for an asynchronous copy);
//   device  dequantisation + inverse DCT (libjpeg's default JDCT_ISLOW, jidctint.c, bit for bit), one thread per 8x8 block;
//           then ONE kernel per image that, for every pixel of the CROP only, evaluates the bilinear taps of
//           cv2.resize(INTER_LINEAR) on the full-resolution image -- each tap's chroma through libjpeg's "fancy" triangle
//           upsampling (jdsample.c) and its integer YCbCr -> RGB tables (jdcolor.c) -- and writes uint8 HWC and / or the
//           normalised fp32 CHW tensor.  No full-resolution RGB image is ever materialised.
// Results are bit-identical to libjpeg-turbo + the restated cv2.resize arithmetic (oracle/reader.py, tests/test_reader.py).
#include <cstring>
#include <vector>

#include "common.h"

namespace endo {

// ---------------------------------------------------------------------------------------------
// host: parser and entropy decoder
// ---------------------------------------------------------------------------------------------
static const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huffman {
    bool present = false;
    uint8_t bits[17] = {0};
    uint8_t values[256] = {0};
    // canonical code tables (ITU T.81 F.2.2.3): per length the smallest code, its value index and the largest code
    int32_t mincode[17], maxcode[18], valptr[17];
    uint8_t look_len[512];          // 9-bit prefix -> code length (0: longer than 9 bits) and symbol
    uint8_t look_sym[512];
    void build() {
        int code = 0, k = 0;
        for (int l = 1; l <= 16; ++l) {
            valptr[l] = k;
            mincode[l] = code;
            code += bits[l];
            k += bits[l];
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        std::memset(look_len, 0, sizeof(look_len));
        int c = 0, idx = 0;
        for (int l = 1; l <= 9; ++l) {
            for (int i = 0; i < bits[l]; ++i, ++idx, ++c) {
                const int first = c << (9 - l);
                for (int f = 0; f < (1 << (9 - l)) && first + f < 512; ++f) { look_len[first + f] = static_cast<uint8_t>(l); look_sym[first + f] = values[idx]; }
            }
            c <<= 1;
        }
    }
};

struct JpegHeader {
    int width = 0, height = 0, ncomp = 0;
    int h[3] = {1, 1, 1}, v[3] = {1, 1, 1}, tq[3] = {0, 0, 0}, td[3] = {0, 0, 0}, ta[3] = {0, 0, 0};
    int hmax = 1, vmax = 1, mcus_x = 0, mcus_y = 0;
    int blocks_w[3] = {0, 0, 0}, blocks_h[3] = {0, 0, 0};
    int64_t block_off[3] = {0, 0, 0}, total_blocks = 0;
    int restart = 0;
    uint16_t quant[4][64];          // natural (row-major) order
    bool quant_present[4] = {false, false, false, false};
    Huffman dc[4], ac[4];
    const uint8_t* scan = nullptr;
    int64_t scan_len = 0;
};

static int parse_jpeg(const uint8_t* d, int64_t n, JpegHeader& hd) {
    if (!d || n < 4 || d[0] != 0xFF || d[1] != 0xD8) return ENDO_E_BADARG;
    int64_t p = 2;
    bool have_frame = false;
    while (p + 4 <= n) {
        if (d[p] != 0xFF) return ENDO_E_BADARG;
        while (p < n && d[p] == 0xFF) ++p;          // fill bytes
        if (p >= n) return ENDO_E_BADARG;
        const int marker = d[p++];
        if (marker == 0xD8 || (marker >= 0xD0 && marker <= 0xD7) || marker == 0x01) continue;
        if (marker == 0xD9) break;
        if (p + 2 > n) return ENDO_E_BADARG;
        const int len = (d[p] << 8) | d[p + 1];
        if (len < 2 || p + len > n) return ENDO_E_BADARG;
        const uint8_t* s = d + p + 2;
        const int body = len - 2;
        if (marker == 0xDB) {          // DQT
            int q = 0;
            while (q < body) {
                const int pq = s[q] >> 4, tq = s[q] & 15;
                ++q;
                if (tq > 3 || q + (pq ? 128 : 64) > body) return ENDO_E_BADARG;
                for (int i = 0; i < 64; ++i) {
                    const int val = pq ? ((s[q] << 8) | s[q + 1]) : s[q];
                    q += pq ? 2 : 1;
                    hd.quant[tq][kZigzag[i]] = static_cast<uint16_t>(val);
                }
                hd.quant_present[tq] = true;
            }
        } else if (marker == 0xC4) {          // DHT
            int q = 0;
            while (q < body) {
                if (q + 17 > body) return ENDO_E_BADARG;
                const int tc = s[q] >> 4, th = s[q] & 15;
                if (tc > 1 || th > 3) return ENDO_E_BADARG;
                Huffman& t = tc ? hd.ac[th] : hd.dc[th];
                int count = 0;
                for (int l = 1; l <= 16; ++l) { t.bits[l] = s[q + l]; count += t.bits[l]; }
                q += 17;
                if (count > 256 || q + count > body) return ENDO_E_BADARG;
                {   // Kraft check: an over-subscribed length table would index the 9-bit lookup (and the canonical codes) out of range
                    int code = 0;
                    for (int l = 1; l <= 16; ++l) {
                        code += t.bits[l];
                        if (code > (1 << l)) return ENDO_E_BADARG;
                        code <<= 1;
                    }
                }
                std::memcpy(t.values, s + q, count);
                q += count;
                t.present = true;
                t.build();
            }
        } else if (marker == 0xC0 || marker == 0xC1) {          // SOF0 / SOF1: sequential Huffman, 8-bit
            if (body < 6 || s[0] != 8) return ENDO_E_UNSUPPORTED;
            hd.height = (s[1] << 8) | s[2];
            hd.width = (s[3] << 8) | s[4];
            hd.ncomp = s[5];
            if ((hd.ncomp != 1 && hd.ncomp != 3) || body < 6 + 3 * hd.ncomp || hd.width <= 0 || hd.height <= 0) return ENDO_E_UNSUPPORTED;
            for (int c = 0; c < hd.ncomp; ++c) {
                hd.h[c] = s[7 + 3 * c] >> 4;
                hd.v[c] = s[7 + 3 * c] & 15;
                hd.tq[c] = s[8 + 3 * c];
                if (hd.h[c] < 1 || hd.h[c] > 2 || hd.v[c] < 1 || hd.v[c] > 2 || hd.tq[c] > 3) return ENDO_E_UNSUPPORTED;
            }
            have_frame = true;
        } else if (marker == 0xC2 || (marker >= 0xC3 && marker <= 0xCF && marker != 0xC4 && marker != 0xC8 && marker != 0xCC)) {
            return ENDO_E_UNSUPPORTED;          // progressive, lossless, arithmetic coding
        } else if (marker == 0xDD) {
            if (body < 2) return ENDO_E_BADARG;
            hd.restart = (s[0] << 8) | s[1];
        } else if (marker == 0xDA) {          // SOS
            if (!have_frame || body < 1 || s[0] != hd.ncomp || body < 1 + 2 * hd.ncomp + 3) return ENDO_E_UNSUPPORTED;   // one interleaved scan
            for (int c = 0; c < hd.ncomp; ++c) {
                hd.td[c] = s[2 + 2 * c] >> 4;
                hd.ta[c] = s[2 + 2 * c] & 15;
                if (hd.td[c] > 3 || hd.ta[c] > 3) return ENDO_E_BADARG;
            }
            hd.scan = d + p + len;
            hd.scan_len = n - (p + len);
            break;
        }
        p += len;
    }
    if (!have_frame || !hd.scan) return ENDO_E_BADARG;
    if (hd.ncomp == 1) { hd.h[0] = 1; hd.v[0] = 1; }          // a single-component scan is not interleaved: one block per MCU
    // chroma layouts this reader knows: 4:4:4, 4:2:2 (h2v1), 4:2:0 (h2v2)
    if (hd.ncomp == 3) {
        if (hd.h[1] != 1 || hd.v[1] != 1 || hd.h[2] != 1 || hd.v[2] != 1) return ENDO_E_UNSUPPORTED;
        if (hd.h[0] == 1 && hd.v[0] == 2) return ENDO_E_UNSUPPORTED;
    }
    for (int c = 0; c < hd.ncomp; ++c) {
        hd.hmax = hd.h[c] > hd.hmax ? hd.h[c] : hd.hmax;
        hd.vmax = hd.v[c] > hd.vmax ? hd.v[c] : hd.vmax;
        if (!hd.quant_present[hd.tq[c]] || !hd.dc[hd.td[c]].present || !hd.ac[hd.ta[c]].present) return ENDO_E_BADARG;
    }
    hd.mcus_x = (hd.width + 8 * hd.hmax - 1) / (8 * hd.hmax);
    hd.mcus_y = (hd.height + 8 * hd.vmax - 1) / (8 * hd.vmax);
    int64_t off = 0;
    for (int c = 0; c < hd.ncomp; ++c) {
        hd.blocks_w[c] = hd.mcus_x * hd.h[c];
        hd.blocks_h[c] = hd.mcus_y * hd.v[c];
        hd.block_off[c] = off;
        off += static_cast<int64_t>(hd.blocks_w[c]) * hd.blocks_h[c];
    }
    hd.total_blocks = off;
    return 0;
}

struct BitReader {
    const uint8_t* p;
    const uint8_t* end;
    uint64_t acc = 0;
    int count = 0;
    bool marker_hit = false;
    void fill() {
        while (count <= 56) {
            int byte = 0;
            if (!marker_hit && p < end) {
                byte = *p;
                if (byte == 0xFF) {
                    if (p + 1 < end && p[1] == 0x00) p += 2;
                    else { marker_hit = true; byte = 0; }          // a marker: feed zeros until the caller consumes it
                } else {
                    ++p;
                }
            }
            acc |= static_cast<uint64_t>(byte) << (56 - count);
            count += 8;
        }
    }
    inline int peek(int nbits) { return static_cast<int>(acc >> (64 - nbits)); }
    inline void skip(int nbits) { acc <<= nbits; count -= nbits; }
    inline int get(int nbits) {
        if (nbits == 0) return 0;
        const int v = peek(nbits);
        skip(nbits);
        return v;
    }
    // Byte-align, drop the buffered bits (fill() never reads past a marker, so what is buffered is the interval's padding) and step
    // over the RSTn marker -- whether or not fill() had already run into it: an interval whose last bits were consumed without
    // another fill() leaves p AT the marker with marker_hit still clear.  FF fill bytes in front of the marker are legal (T.81 B.1.1.2).
    // false: a restart is due and no RSTn follows (damaged stream).
    bool reset_at_restart() {
        acc = 0;
        count = 0;
        marker_hit = false;
        while (p + 1 < end && p[0] == 0xFF && p[1] == 0xFF) ++p;
        if (p + 1 < end && p[0] == 0xFF && p[1] >= 0xD0 && p[1] <= 0xD7) { p += 2; return true; }
        return false;
    }
};

static inline int decode_symbol(BitReader& br, const Huffman& t) {
    if (br.count < 16) br.fill();
    const int look = br.peek(9);
    const int l = t.look_len[look];
    if (l) { br.skip(l); return t.look_sym[look]; }
    int code = br.peek(10), len = 10;
    while (len <= 16 && code > t.maxcode[len]) { ++len; code = br.peek(len); }
    if (len > 16) return -1;
    br.skip(len);
    return t.values[t.valptr[len] + code - t.mincode[len]];
}

static inline int extend(int v, int nbits) { return v < (1 << (nbits - 1)) ? v - (1 << nbits) + 1 : v; }

// blocks: total_blocks * 64 int16, component planes one after the other, inside a plane row-major blocks of 64 coefficients
// in natural order.  Returns 0, or a negative code for a damaged stream.
static int entropy_decode(const JpegHeader& hd, int16_t* blocks) {
    std::memset(blocks, 0, static_cast<size_t>(hd.total_blocks) * 64 * sizeof(int16_t));
    BitReader br{hd.scan, hd.scan + hd.scan_len};
    int pred[3] = {0, 0, 0};
    int until_restart = hd.restart;
    for (int my = 0; my < hd.mcus_y; ++my) {
        for (int mx = 0; mx < hd.mcus_x; ++mx) {
            if (hd.restart && until_restart == 0) {
                if (!br.reset_at_restart()) return ENDO_E_BADARG;
                pred[0] = pred[1] = pred[2] = 0;
                until_restart = hd.restart;
            }
            for (int c = 0; c < hd.ncomp; ++c) {
                const Huffman& dct = hd.dc[hd.td[c]];
                const Huffman& act = hd.ac[hd.ta[c]];
                for (int by = 0; by < hd.v[c]; ++by)
                    for (int bx = 0; bx < hd.h[c]; ++bx) {
                        int16_t* blk = blocks + (hd.block_off[c] + static_cast<int64_t>(my * hd.v[c] + by) * hd.blocks_w[c] + mx * hd.h[c] + bx) * 64;
                        const int s = decode_symbol(br, dct);
                        if (s < 0 || s > 11) return ENDO_E_BADARG;
                        if (s) {
                            if (br.count < s) br.fill();
                            pred[c] += extend(br.get(s), s);
                        }
                        blk[0] = static_cast<int16_t>(pred[c]);
                        for (int k = 1; k < 64;) {
                            const int rs = decode_symbol(br, act);
                            if (rs < 0) return ENDO_E_BADARG;
                            const int r = rs >> 4, sz = rs & 15;
                            if (sz == 0) {
                                if (r != 15) break;          // end of block
                                k += 16;
                                continue;
                            }
                            k += r;
                            if (k > 63) return ENDO_E_BADARG;
                            if (br.count < sz) br.fill();
                            blk[kZigzag[k]] = static_cast<int16_t>(extend(br.get(sz), sz));
                            ++k;
                        }
                    }
            }
            --until_restart;
        }
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// device
// ---------------------------------------------------------------------------------------------
struct JpegPlanes {
    int width, height, ncomp;
    int hs, vs;                       // luma sampling factors (chroma is 1 x 1): 1 / 2
    int stride[3];                    // bytes per row of each component's plane (blocks_w * 8)
    int cw, ch;                       // REAL chroma size: ceil(width / hs), ceil(height / vs)
    int64_t plane_off[3];             // byte offsets of the planes in the workspace
    int64_t block_off[3];
    int blocks_w[3];
    int64_t total_blocks;
};

// jidctint.c (jpeg_idct_islow), one dimension.  CONST_BITS = 13, PASS1_BITS = 2.
__device__ __forceinline__ void idct_islow_1d(const int (&v)[8], int (&o)[8], int shift) {
    int z2 = v[2], z3 = v[6];
    int z1 = (z2 + z3) * 4433;
    int tmp2 = z1 + z3 * (-15137);
    int tmp3 = z1 + z2 * 6270;
    z2 = v[0]; z3 = v[4];
    int tmp0 = (z2 + z3) << 13;
    int tmp1 = (z2 - z3) << 13;
    const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = v[7]; tmp1 = v[5]; tmp2 = v[3]; tmp3 = v[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    int z4 = tmp1 + tmp3;
    const int z5 = (z3 + z4) * 9633;
    tmp0 *= 2446; tmp1 *= 16819; tmp2 *= 25172; tmp3 *= 12299;
    z1 *= -7373; z2 *= -20995; z3 *= -16069; z4 *= -3196;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    const int rnd = 1 << (shift - 1);
    o[0] = (tmp10 + tmp3 + rnd) >> shift; o[7] = (tmp10 - tmp3 + rnd) >> shift;
    o[1] = (tmp11 + tmp2 + rnd) >> shift; o[6] = (tmp11 - tmp2 + rnd) >> shift;
    o[2] = (tmp12 + tmp1 + rnd) >> shift; o[5] = (tmp12 - tmp1 + rnd) >> shift;
    o[3] = (tmp13 + tmp0 + rnd) >> shift; o[4] = (tmp13 - tmp0 + rnd) >> shift;
}

// libjpeg's IDCT range-limit table (jdmaster.c prepare_range_limit_table), index masked to 10 bits, centre offset included
__device__ __forceinline__ uint8_t idct_range_limit(int x) {
    const int i = x & 1023;
    return static_cast<uint8_t>(i < 128 ? i + 128 : i < 512 ? 255 : i < 896 ? 0 : i - 896);
}

// one thread per 8x8 block: dequantise, both passes in registers, 8 x 8 bytes out
__global__ void __launch_bounds__(64) jpeg_idct_kernel(const int16_t* __restrict__ coef, const uint16_t* __restrict__ quant, uint8_t* __restrict__ ws,
                                                      const JpegPlanes pl) {
    const int64_t b = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (b >= pl.total_blocks) return;
    const int c = (pl.ncomp == 3 && b >= pl.block_off[1]) ? (b >= pl.block_off[2] ? 2 : 1) : 0;
    const int64_t local = b - pl.block_off[c];
    const int by = static_cast<int>(local / pl.blocks_w[c]), bx = static_cast<int>(local - static_cast<int64_t>(by) * pl.blocks_w[c]);
    const int16_t* src = coef + b * 64;
    const uint16_t* q = quant + c * 64;
    int w[8][8];
#pragma unroll
    for (int col = 0; col < 8; ++col) {
        int v[8], o[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = static_cast<int>(src[k * 8 + col]) * static_cast<int>(q[k * 8 + col]);
        idct_islow_1d(v, o, 13 - 2);
#pragma unroll
        for (int k = 0; k < 8; ++k) w[k][col] = o[k];
    }
    uint8_t* dst = ws + pl.plane_off[c] + static_cast<int64_t>(by) * 8 * pl.stride[c] + bx * 8;
#pragma unroll
    for (int row = 0; row < 8; ++row) {
        int o[8];
        idct_islow_1d(w[row], o, 13 + 2 + 3);
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            lo |= static_cast<uint32_t>(idct_range_limit(o[k])) << (8 * k);
            hi |= static_cast<uint32_t>(idct_range_limit(o[4 + k])) << (8 * k);
        }
        *reinterpret_cast<uint2*>(dst + static_cast<int64_t>(row) * pl.stride[c]) = make_uint2(lo, hi);
    }
}

// jdsample.c: "fancy" (triangle filter) chroma upsampling, evaluated at one full-resolution position
__device__ __forceinline__ int chroma_at(const uint8_t* __restrict__ p, int stride, int cw, int ch, int hs, int vs, int yy, int xx) {
    if (hs == 1) return p[static_cast<int64_t>(yy) * stride + xx];          // 4:4:4
    const int cx = xx >> 1, odd = xx & 1;
    if (vs == 1) {          // h2v1_fancy_upsample
        const uint8_t* r = p + static_cast<int64_t>(yy) * stride;
        const int t = r[cx];
        if (!odd) return cx == 0 ? t : (3 * t + r[cx - 1] + 1) >> 2;
        return cx == cw - 1 ? t : (3 * t + r[cx + 1] + 2) >> 2;
    }
    // h2v2_fancy_upsample: 3/4 nearer row + 1/4 further row, then the same horizontally; edges replicate
    const int cy = yy >> 1;
    const int other = (yy & 1) ? min(cy + 1, ch - 1) : max(cy - 1, 0);
    const uint8_t* r0 = p + static_cast<int64_t>(cy) * stride;
    const uint8_t* r1 = p + static_cast<int64_t>(other) * stride;
    const int t = 3 * r0[cx] + r1[cx];
    if (!odd) return cx == 0 ? (t * 4 + 8) >> 4 : (t * 3 + 3 * r0[cx - 1] + r1[cx - 1] + 8) >> 4;
    return cx == cw - 1 ? (t * 4 + 7) >> 4 : (t * 3 + 3 * r0[cx + 1] + r1[cx + 1] + 7) >> 4;
}

__device__ __forceinline__ int clamp255(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }

// jdcolor.c ycc_rgb_convert with the tables of build_ycc_rgb_table written out (SCALEBITS 16)
__device__ __forceinline__ void ycc_to_rgb(int y, int cb, int cr, int (&rgb)[3]) {
    const int xb = cb - 128, xr = cr - 128;
    rgb[0] = clamp255(y + ((91881 * xr + 32768) >> 16));
    rgb[1] = clamp255(y + ((-22554 * xb + 32768 - 46802 * xr) >> 16));
    rgb[2] = clamp255(y + ((116130 * xb + 32768) >> 16));
}

// resize.cpp (INTER_LINEAR, 8-bit): tap positions and 11-bit coefficients of one destination coordinate
__device__ __forceinline__ void linear_tap(int d, int src_size, double scale, int& s0, int& s1, int& c0, int& c1) {
    float f = static_cast<float>((d + 0.5) * scale - 0.5);
    int s = static_cast<int>(floorf(f));
    f -= static_cast<float>(s);
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= src_size - 1) { f = 0.f; s = src_size - 1; }
    s0 = s;
    s1 = min(s + 1, src_size - 1);
    c0 = static_cast<int>(rintf((1.f - f) * 2048.f));
    c1 = static_cast<int>(rintf(f * 2048.f));
}

// one thread per pixel of the crop
__global__ void __launch_bounds__(256) jpeg_resize_crop_kernel(const uint8_t* __restrict__ ws, const JpegPlanes pl, double scale, int start_h, int start_w,
                                                             int out_h, int out_w, int rgb_order, uint8_t* __restrict__ out_hwc,
                                                             float* __restrict__ out_chw) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= out_w) return;
    int ys[2], xs[2], b[2], a[2];
    linear_tap(y + start_h, pl.height, scale, ys[0], ys[1], b[0], b[1]);
    linear_tap(x + start_w, pl.width, scale, xs[0], xs[1], a[0], a[1]);
    const uint8_t* py = ws + pl.plane_off[0];
    const uint8_t* pcb = ws + pl.plane_off[1];
    const uint8_t* pcr = ws + pl.plane_off[2];
    int px[2][2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int lum = py[static_cast<int64_t>(ys[i]) * pl.stride[0] + xs[j]];
            if (pl.ncomp == 1) {
                px[i][j][0] = px[i][j][1] = px[i][j][2] = lum;
            } else {
                const int cb = chroma_at(pcb, pl.stride[1], pl.cw, pl.ch, pl.hs, pl.vs, ys[i], xs[j]);
                const int cr = chroma_at(pcr, pl.stride[2], pl.cw, pl.ch, pl.hs, pl.vs, ys[i], xs[j]);
                ycc_to_rgb(lum, cb, cr, px[i][j]);
            }
        }
    const int64_t plane = static_cast<int64_t>(out_h) * out_w;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int s0 = px[0][0][c] * a[0] + px[0][1][c] * a[1];
        const int s1 = px[1][0][c] * a[0] + px[1][1][c] * a[1];
        int v = (((b[0] * (s0 >> 4)) >> 16) + ((b[1] * (s1 >> 4)) >> 16) + 2) >> 2;
        v = clamp255(v);
        const int oc = rgb_order ? c : 2 - c;
        if (out_hwc) out_hwc[(static_cast<int64_t>(y) * out_w + x) * 3 + oc] = static_cast<uint8_t>(v);
        // albumentations Normalize(mean 0.5, std 0.5, max_pixel_value 255) in fp32: (v - 127.5) * float32(1 / 127.5)
        if (out_chw) out_chw[oc * plane + static_cast<int64_t>(y) * out_w + x] = (static_cast<float>(v) - 127.5f) * (1.0f / 127.5f);
    }
}

static int64_t align256(int64_t v) { return (v + 255) & ~static_cast<int64_t>(255); }

struct JpegLayout {
    int64_t coef_bytes, quant_off, planes_off, total;
    JpegPlanes pl;
};

static JpegLayout jpeg_layout(const JpegHeader& hd) {
    JpegLayout l{};
    l.coef_bytes = hd.total_blocks * 64 * static_cast<int64_t>(sizeof(int16_t));
    l.quant_off = align256(l.coef_bytes);
    l.planes_off = align256(l.quant_off + 3 * 64 * static_cast<int64_t>(sizeof(uint16_t)));
    JpegPlanes& pl = l.pl;
    pl.width = hd.width; pl.height = hd.height; pl.ncomp = hd.ncomp;
    pl.hs = hd.hmax; pl.vs = hd.vmax;
    pl.cw = (hd.width + hd.hmax - 1) / hd.hmax;
    pl.ch = (hd.height + hd.vmax - 1) / hd.vmax;
    pl.total_blocks = hd.total_blocks;
    int64_t off = l.planes_off;
    for (int c = 0; c < 3; ++c) {
        const int cc = c < hd.ncomp ? c : 0;
        pl.stride[c] = hd.blocks_w[cc] * 8;
        pl.blocks_w[c] = hd.blocks_w[cc];
        pl.block_off[c] = hd.block_off[cc];
        pl.plane_off[c] = c < hd.ncomp ? off : pl.plane_off[0];
        if (c < hd.ncomp) off += align256(static_cast<int64_t>(hd.blocks_w[c]) * 8 * hd.blocks_h[c] * 8);
    }
    l.total = off;
    return l;
}

}  // namespace endo

using namespace endo;

extern "C" int endo_jpeg_info(const uint8_t* data, int64_t size, int32_t* info) {
    if (!info) return ENDO_E_BADARG;
    JpegHeader hd;
    const int rc = parse_jpeg(data, size, hd);
    if (rc) return rc;
    info[0] = hd.width; info[1] = hd.height; info[2] = hd.ncomp; info[3] = hd.hmax; info[4] = hd.vmax;
    info[5] = hd.mcus_x; info[6] = hd.mcus_y;
    for (int c = 0; c < 3; ++c) { info[7 + 2 * c] = c < hd.ncomp ? hd.blocks_w[c] : 0; info[8 + 2 * c] = c < hd.ncomp ? hd.blocks_h[c] : 0; }
    info[13] = static_cast<int32_t>(hd.total_blocks);
    info[14] = hd.restart;
    info[15] = 0;
    return 0;
}

extern "C" int endo_jpeg_entropy_decode(const uint8_t* data, int64_t size, int16_t* blocks, int64_t capacity_blocks, uint16_t* quant) {
    if (!blocks || !quant) return ENDO_E_BADARG;
    JpegHeader hd;
    int rc = parse_jpeg(data, size, hd);
    if (rc) return rc;
    if (capacity_blocks < hd.total_blocks) return ENDO_E_BADARG;
    for (int c = 0; c < 3; ++c) std::memcpy(quant + 64 * c, hd.quant[hd.tq[c < hd.ncomp ? c : 0]], 64 * sizeof(uint16_t));
    return entropy_decode(hd, blocks);
}

extern "C" int64_t endo_jpeg_workspace_bytes(const uint8_t* data, int64_t size) {
    JpegHeader hd;
    if (parse_jpeg(data, size, hd)) return -1;
    return jpeg_layout(hd).total;
}

extern "C" int endo_jpeg_decode_crop(const uint8_t* data, int64_t size, double downsampling, int start_h, int end_h, int start_w, int end_w,
                                     int rgb_order, uint8_t* out_hwc, float* out_chw, void* staging, void* workspace, int64_t workspace_bytes,
                                     void* stream_) {
    if (!staging || !workspace || (!out_hwc && !out_chw) || downsampling <= 0.0) return ENDO_E_BADARG;
    JpegHeader hd;
    int rc = parse_jpeg(data, size, hd);
    if (rc) return rc;
    const JpegLayout l = jpeg_layout(hd);
    if (workspace_bytes < l.total) return ENDO_E_BADARG;
    // cv::resize(src, dst, Size(), fx, fy): dsize = saturate_cast<int>(ssize * f), scale = 1 / f
    const double inv = 1.0 / downsampling;
    const int dst_w = static_cast<int>(nearbyint(hd.width * inv)), dst_h = static_cast<int>(nearbyint(hd.height * inv));
    if (start_h < 0 || start_w < 0 || end_h > dst_h || end_w > dst_w || end_h <= start_h || end_w <= start_w) return ENDO_E_BADARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    char* host = static_cast<char*>(staging);
    rc = entropy_decode(hd, reinterpret_cast<int16_t*>(host));
    if (rc) return rc;
    uint16_t* hq = reinterpret_cast<uint16_t*>(host + l.quant_off);
    for (int c = 0; c < 3; ++c) std::memcpy(hq + 64 * c, hd.quant[hd.tq[c < hd.ncomp ? c : 0]], 64 * sizeof(uint16_t));
    char* dev = static_cast<char*>(workspace);
    ENDO_CHECK(hipMemcpyAsync(dev, host, static_cast<size_t>(l.planes_off), hipMemcpyHostToDevice, stream));
    ProfScope prof(kProfSmall, stream, 0.0, 0.0);
    jpeg_idct_kernel<<<static_cast<unsigned>((hd.total_blocks + 63) / 64), 64, 0, stream>>>(
        reinterpret_cast<const int16_t*>(dev), reinterpret_cast<const uint16_t*>(dev + l.quant_off), reinterpret_cast<uint8_t*>(dev), l.pl);
    const int out_h = end_h - start_h, out_w = end_w - start_w;
    jpeg_resize_crop_kernel<<<dim3((out_w + 255) / 256, out_h), 256, 0, stream>>>(reinterpret_cast<const uint8_t*>(dev), l.pl, 1.0 / inv, start_h, start_w,
                                                                                  out_h, out_w, rgb_order, out_hwc, out_chw);
    ENDO_LAUNCH_CHECK();
    return 0;
}
