#include "image_io.h"
#include <zlib.h>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <algorithm>
#include <cmath>

namespace lrt {

static std::vector<uint8_t> read_file(const std::string &path) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open \"" + path + "\"");
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> d((size_t) n);
    if (n && fread(d.data(), 1, (size_t) n, f) != (size_t) n) { fclose(f); throw std::runtime_error("short read on \"" + path + "\""); }
    fclose(f);
    return d;
}

static std::vector<uint8_t> zlib_inflate(const uint8_t *src, size_t n, size_t expected) {
    std::vector<uint8_t> out(expected);
    uLongf dl = (uLongf) expected;
    int rc = uncompress(out.data(), &dl, src, (uLong) n);
    if (rc != Z_OK) throw std::runtime_error("zlib: inflate failed (" + std::to_string(rc) + ")");
    out.resize(dl);
    return out;
}

// ------------------------------------------------------------------ PNG
static uint32_t be32(const uint8_t *p) { return ((uint32_t) p[0] << 24) | ((uint32_t) p[1] << 16) | ((uint32_t) p[2] << 8) | p[3]; }

Image read_png(const std::string &path) {
    std::vector<uint8_t> f = read_file(path);
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n' };
    if (f.size() < 8 || memcmp(f.data(), sig, 8)) throw std::runtime_error("\"" + path + "\": not a PNG file");
    size_t i = 8; uint32_t w = 0, h = 0; int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat;
    while (i + 12 <= f.size()) {
        uint32_t len = be32(&f[i]); const uint8_t *type = &f[i + 4], *data = &f[i + 8];
        if (i + 12 + len > f.size()) throw std::runtime_error("\"" + path + "\": truncated PNG chunk");
        if (!memcmp(type, "IHDR", 4)) { w = be32(data); h = be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12]; }
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
        else if (!memcmp(type, "IEND", 4)) break;
        i += 12 + len;
    }
    if (!w || !h) throw std::runtime_error("\"" + path + "\": missing IHDR");
    if (interlace) throw std::runtime_error("\"" + path + "\": interlaced PNG is not supported");
    if (depth != 8 && depth != 16) throw std::runtime_error("\"" + path + "\": only 8/16-bit PNG is supported");
    int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!ch) throw std::runtime_error("\"" + path + "\": palette PNG is not supported");
    size_t bpp = (size_t) ch * depth / 8, stride = bpp * w;
    std::vector<uint8_t> raw = zlib_inflate(idat.data(), idat.size(), (stride + 1) * h);
    if (raw.size() != (stride + 1) * h) throw std::runtime_error("\"" + path + "\": unexpected PNG data size");
    std::vector<uint8_t> img(stride * h);
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t *in = &raw[(stride + 1) * y]; uint8_t ft = *in++;
        uint8_t *out = &img[stride * y]; const uint8_t *up = y ? &img[stride * (y - 1)] : nullptr;
        for (size_t x = 0; x < stride; ++x) {
            int a = x >= bpp ? out[x - bpp] : 0, b = up ? up[x] : 0, c = (up && x >= bpp) ? up[x - bpp] : 0, v = in[x];
            switch (ft) {
                case 0: break;
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) >> 1; break;
                case 4: { int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
                          v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
                default: throw std::runtime_error("\"" + path + "\": bad PNG filter type");
            }
            out[x] = (uint8_t) v;
        }
    }
    Image r; r.width = (int) w; r.height = (int) h; r.channels = ch; r.bits_per_channel = depth; r.srgb = true;
    r.data.resize((size_t) w * h * ch);
    for (size_t k = 0; k < r.data.size(); ++k)
        r.data[k] = depth == 8 ? img[k] * (1.f / 255.f) : (float) ((img[2 * k] << 8) | img[2 * k + 1]) * (1.f / 65535.f);
    return r;
}

// ------------------------------------------------------------------ EXR
static float half_to_float(uint16_t h) {
    uint32_t s = (uint32_t) (h >> 15) << 31, e = (h >> 10) & 31, m = h & 1023, u;
    if (e == 0) {
        if (m == 0) u = s;
        else { int sh = 0; while (!(m & 1024)) { m <<= 1; ++sh; } u = s | ((uint32_t) (113 - sh) << 23) | ((m & 1023) << 13); }
    } else if (e == 31) u = s | 0x7f800000u | (m << 13);
    else u = s | ((e + 112) << 23) | (m << 13);
    float f; memcpy(&f, &u, 4); return f;
}

namespace piz {
// Restated from the published OpenEXR PIZ scheme (ImfPizCompressor / ImfHuf /
// ImfWav): 16-bit wavelet transform + canonical Huffman coding + value LUT.
const int USHORT_RANGE = 1 << 16, BITMAP_SIZE = USHORT_RANGE >> 3;
const int HUF_ENCBITS = 16, HUF_DECBITS = 14, HUF_ENCSIZE = (1 << HUF_ENCBITS) + 1, HUF_DECSIZE = 1 << HUF_DECBITS, HUF_DECMASK = HUF_DECSIZE - 1;
const int SHORT_ZEROCODE_RUN = 59, LONG_ZEROCODE_RUN = 63, SHORTEST_LONG_RUN = 2 + LONG_ZEROCODE_RUN - SHORT_ZEROCODE_RUN;

struct HufDec { int len = 0; int lit = 0; std::vector<int> p; };

static inline int64_t get_bits(int n, int64_t &c, int &lc, const uint8_t *&in) {
    while (lc < n) { c = (c << 8) | *in++; lc += 8; }
    lc -= n; return (c >> lc) & ((1 << n) - 1);
}
static void canonical_table(std::vector<int64_t> &hcode) {
    int64_t n[59]; for (auto &x : n) x = 0;
    for (int i = 0; i < HUF_ENCSIZE; ++i) n[hcode[i]] += 1;
    int64_t c = 0;
    for (int i = 58; i > 0; --i) { int64_t nc = (c + n[i]) >> 1; n[i] = c; c = nc; }
    for (int i = 0; i < HUF_ENCSIZE; ++i) { int l = (int) hcode[i]; if (l > 0) hcode[i] = l | (n[l]++ << 6); }
}
static void unpack_enc_table(const uint8_t *&p, const uint8_t *end, int im, int iM, std::vector<int64_t> &hcode) {
    std::fill(hcode.begin(), hcode.end(), 0);
    int64_t c = 0; int lc = 0;
    for (; im <= iM; im++) {
        if (p > end) throw std::runtime_error("PIZ: truncated Huffman table");
        int64_t l = hcode[im] = get_bits(6, c, lc, p);
        if (l == LONG_ZEROCODE_RUN) {
            int zerun = (int) get_bits(8, c, lc, p) + SHORTEST_LONG_RUN;
            if (im + zerun > iM + 1) throw std::runtime_error("PIZ: Huffman table overrun");
            while (zerun--) hcode[im++] = 0;
            im--;
        } else if (l >= SHORT_ZEROCODE_RUN) {
            int zerun = (int) l - SHORT_ZEROCODE_RUN + 2;
            if (im + zerun > iM + 1) throw std::runtime_error("PIZ: Huffman table overrun");
            while (zerun--) hcode[im++] = 0;
            im--;
        }
    }
    canonical_table(hcode);
}
static void build_dec_table(const std::vector<int64_t> &hcode, int im, int iM, std::vector<HufDec> &hdec) {
    for (; im <= iM; im++) {
        int64_t c = hcode[im] >> 6; int l = (int) (hcode[im] & 63);
        if (c >> l) throw std::runtime_error("PIZ: invalid Huffman code");
        if (l > HUF_DECBITS) {
            HufDec &pl = hdec[(size_t) (c >> (l - HUF_DECBITS))];
            if (pl.len) throw std::runtime_error("PIZ: invalid Huffman table");
            pl.lit++; pl.p.push_back(im);
        } else if (l) {
            size_t base = (size_t) (c << (HUF_DECBITS - l));
            for (int64_t i = (int64_t) 1 << (HUF_DECBITS - l); i > 0; i--, base++) {
                HufDec &pl = hdec[base];
                if (pl.len || !pl.p.empty()) throw std::runtime_error("PIZ: invalid Huffman table");
                pl.len = l; pl.lit = im;
            }
        }
    }
}
static void huf_decode(const std::vector<int64_t> &hcode, const std::vector<HufDec> &hdec, const uint8_t *in, int ni, int rlc, int no, uint16_t *out) {
    int64_t c = 0; int lc = 0; uint16_t *outb = out, *oe = out + no; const uint8_t *ie = in + (ni + 7) / 8;
    auto get_code = [&](int po) {
        if (po == rlc) {
            if (lc < 8) { c = (c << 8) | *in++; lc += 8; }
            lc -= 8; int cs = (int) ((c >> lc) & 0xff);
            if (out + cs > oe || out == outb) throw std::runtime_error("PIZ: bad run-length code");
            uint16_t s = out[-1]; while (cs-- > 0) *out++ = s;
        } else if (out < oe) *out++ = (uint16_t) po;
        else throw std::runtime_error("PIZ: output overrun");
    };
    while (in < ie) {
        c = (c << 8) | *in++; lc += 8;
        while (lc >= HUF_DECBITS) {
            const HufDec &pl = hdec[(size_t) ((c >> (lc - HUF_DECBITS)) & HUF_DECMASK)];
            if (pl.len) { lc -= pl.len; get_code(pl.lit); }
            else {
                if (pl.p.empty()) throw std::runtime_error("PIZ: invalid code");
                int j;
                for (j = 0; j < pl.lit; j++) {
                    int l = (int) (hcode[pl.p[j]] & 63);
                    while (lc < l && in < ie) { c = (c << 8) | *in++; lc += 8; }
                    if (lc >= l && (hcode[pl.p[j]] >> 6) == ((c >> (lc - l)) & (((int64_t) 1 << l) - 1))) { lc -= l; get_code(pl.p[j]); break; }
                }
                if (j == pl.lit) throw std::runtime_error("PIZ: invalid code");
            }
        }
    }
    int i = (8 - ni) & 7; c >>= i; lc -= i;
    while (lc > 0) {
        const HufDec &pl = hdec[(size_t) ((c << (HUF_DECBITS - lc)) & HUF_DECMASK)];
        if (pl.len) { lc -= pl.len; get_code(pl.lit); }
        else throw std::runtime_error("PIZ: invalid code");
    }
    if (out - outb != no) throw std::runtime_error("PIZ: not enough data");
}
static void huf_uncompress(const uint8_t *comp, int ncomp, uint16_t *raw, int nraw) {
    if (ncomp == 0) { if (nraw) throw std::runtime_error("PIZ: not enough data"); return; }
    if (ncomp < 20) throw std::runtime_error("PIZ: truncated Huffman header");
    auto rd = [&](int o) { return (int) ((uint32_t) comp[o] | ((uint32_t) comp[o + 1] << 8) | ((uint32_t) comp[o + 2] << 16) | ((uint32_t) comp[o + 3] << 24)); };
    int im = rd(0), iM = rd(4), nbits = rd(12);
    if (im < 0 || im >= HUF_ENCSIZE || iM < 0 || iM >= HUF_ENCSIZE) throw std::runtime_error("PIZ: bad Huffman header");
    const uint8_t *ptr = comp + 20;
    std::vector<int64_t> freq(HUF_ENCSIZE); std::vector<HufDec> hdec(HUF_DECSIZE);
    unpack_enc_table(ptr, comp + ncomp, im, iM, freq);
    if (nbits > 8 * (ncomp - (int) (ptr - comp))) throw std::runtime_error("PIZ: bad bit count");
    build_dec_table(freq, im, iM, hdec);
    huf_decode(freq, hdec, ptr, nbits, iM, nraw, raw);
}
static inline void wdec14(uint16_t l, uint16_t h, uint16_t &a, uint16_t &b) {
    short ls = (short) l, hs = (short) h; int hi = hs, ai = ls + (hi & 1) + (hi >> 1);
    a = (uint16_t) (short) ai; b = (uint16_t) (short) (ai - hi);
}
static inline void wdec16(uint16_t l, uint16_t h, uint16_t &a, uint16_t &b) {
    int m = l, d = h, bb = (m - (d >> 1)) & 0xffff, aa = (d + bb - 0x8000) & 0xffff;
    b = (uint16_t) bb; a = (uint16_t) aa;
}
static void wav2_decode(uint16_t *in, int nx, int ox, int ny, int oy, uint16_t mx) {
    bool w14 = mx < (1 << 14);
    int n = std::min(nx, ny), p = 1, p2;
    while (p <= n) p <<= 1;
    p >>= 1; p2 = p; p >>= 1;
    while (p >= 1) {
        uint16_t *py = in, *ey = in + oy * (ny - p2);
        int oy1 = oy * p, oy2 = oy * p2, ox1 = ox * p, ox2 = ox * p2;
        uint16_t i00, i01, i10, i11;
        for (; py <= ey; py += oy2) {
            uint16_t *px = py, *ex = py + ox * (nx - p2);
            for (; px <= ex; px += ox2) {
                uint16_t *p01 = px + ox1, *p10 = px + oy1, *p11 = p10 + ox1;
                if (w14) { wdec14(*px, *p10, i00, i10); wdec14(*p01, *p11, i01, i11); wdec14(i00, i01, *px, *p01); wdec14(i10, i11, *p10, *p11); }
                else { wdec16(*px, *p10, i00, i10); wdec16(*p01, *p11, i01, i11); wdec16(i00, i01, *px, *p01); wdec16(i10, i11, *p10, *p11); }
            }
            if (nx & p) {
                uint16_t *p10 = px + oy1;
                if (w14) wdec14(*px, *p10, i00, *p10); else wdec16(*px, *p10, i00, *p10);
                *px = i00;
            }
        }
        if (ny & p) {
            uint16_t *px = py, *ex = py + ox * (nx - p2);
            for (; px <= ex; px += ox2) {
                uint16_t *p01 = px + ox1;
                if (w14) wdec14(*px, *p01, i00, *p01); else wdec16(*px, *p01, i00, *p01);
                *px = i00;
            }
        }
        p2 = p; p >>= 1;
    }
}
// Decompress one PIZ block of `ny` scanlines into per-scanline, per-channel uint16 words.
static void decompress(const uint8_t *src, int nsrc, int nx, int ny, const std::vector<int> &chan_words, std::vector<uint16_t> &out) {
    int total = 0; for (int s : chan_words) total += nx * ny * s;
    out.assign((size_t) total, 0);
    if (nsrc == 0) return;
    std::vector<uint8_t> bitmap(BITMAP_SIZE, 0);
    if (nsrc < 4) throw std::runtime_error("PIZ: truncated block");
    int minnz = src[0] | (src[1] << 8), maxnz = src[2] | (src[3] << 8);
    const uint8_t *p = src + 4;
    if (maxnz >= BITMAP_SIZE) throw std::runtime_error("PIZ: bad bitmap range");
    if (minnz <= maxnz) { if (p + (maxnz - minnz + 1) > src + nsrc) throw std::runtime_error("PIZ: truncated bitmap"); memcpy(&bitmap[minnz], p, (size_t) (maxnz - minnz + 1)); p += maxnz - minnz + 1; }
    std::vector<uint16_t> lut(USHORT_RANGE, 0);
    int k = 0;
    for (int i = 0; i < USHORT_RANGE; ++i) if (i == 0 || (bitmap[i >> 3] & (1 << (i & 7)))) lut[k++] = (uint16_t) i;
    uint16_t maxv = (uint16_t) (k - 1);
    if (p + 4 > src + nsrc) throw std::runtime_error("PIZ: truncated block");
    int length = (int) ((uint32_t) p[0] | ((uint32_t) p[1] << 8) | ((uint32_t) p[2] << 16) | ((uint32_t) p[3] << 24)); p += 4;
    if (length < 0 || p + length > src + nsrc) throw std::runtime_error("PIZ: bad compressed length");
    std::vector<uint16_t> tmp((size_t) total);
    huf_uncompress(p, length, tmp.data(), total);
    std::vector<uint16_t *> cstart; uint16_t *q = tmp.data();
    for (int s : chan_words) {
        cstart.push_back(q);
        for (int j = 0; j < s; ++j) wav2_decode(q + j, nx, s, ny, nx * s, maxv);
        q += nx * ny * s;
    }
    for (auto &v : tmp) v = lut[v];
    uint16_t *o = out.data();
    for (int y = 0; y < ny; ++y)
        for (size_t c = 0; c < chan_words.size(); ++c) {
            int n = nx * chan_words[c];
            memcpy(o, cstart[c] + (size_t) y * n, (size_t) n * 2); o += n;
        }
}
} // namespace piz

Image read_exr(const std::string &path) {
    std::vector<uint8_t> f = read_file(path);
    auto need = [&](size_t o, size_t n) { if (o + n > f.size()) throw std::runtime_error("\"" + path + "\": truncated EXR"); };
    need(0, 8);
    if (f[0] != 0x76 || f[1] != 0x2f || f[2] != 0x31 || f[3] != 0x01) throw std::runtime_error("\"" + path + "\": not an OpenEXR file");
    uint32_t version; memcpy(&version, &f[4], 4);
    if (version & 0x200) throw std::runtime_error("\"" + path + "\": tiled EXR is not supported");
    if (version & 0x1800) throw std::runtime_error("\"" + path + "\": multi-part/deep EXR is not supported");
    size_t i = 8;
    struct Chan { std::string name; int type; };
    std::vector<Chan> chans; int compression = -1; int dw[4] = { 0, 0, -1, -1 };
    for (;;) {
        need(i, 1); if (f[i] == 0) { ++i; break; }
        size_t j = i; while (j < f.size() && f[j]) ++j; std::string name((char *) &f[i], j - i);
        size_t k = j + 1; while (k < f.size() && f[k]) ++k; std::string type((char *) &f[j + 1], k - j - 1);
        need(k + 1, 4); uint32_t sz; memcpy(&sz, &f[k + 1], 4); size_t v = k + 5; need(v, sz);
        if (name == "channels") {
            size_t q = v;
            while (f[q]) {
                size_t e = q; while (f[e]) ++e;
                Chan c; c.name.assign((char *) &f[q], e - q); int32_t t; memcpy(&t, &f[e + 1], 4); c.type = t;
                int32_t xs, ys; memcpy(&xs, &f[e + 9], 4); memcpy(&ys, &f[e + 13], 4);
                if (xs != 1 || ys != 1) throw std::runtime_error("\"" + path + "\": subsampled EXR channels are not supported");
                chans.push_back(c); q = e + 17;
            }
        } else if (name == "compression") compression = f[v];
        else if (name == "dataWindow") memcpy(dw, &f[v], 16);
        i = v + sz;
    }
    int w = dw[2] - dw[0] + 1, h = dw[3] - dw[1] + 1;
    if (w <= 0 || h <= 0 || chans.empty()) throw std::runtime_error("\"" + path + "\": bad EXR header");
    int lines_per_block = compression == 0 || compression == 1 || compression == 2 ? 1 : compression == 3 ? 16 : compression == 4 ? 32 : 0;
    if (!lines_per_block) throw std::runtime_error("\"" + path + "\": unsupported EXR compression " + std::to_string(compression));
    if (compression == 1) throw std::runtime_error("\"" + path + "\": RLE EXR compression is not supported");
    std::vector<int> cbytes, cwords; size_t line_bytes = 0;
    for (auto &c : chans) { if (c.type < 0 || c.type > 2) throw std::runtime_error("bad EXR channel type"); int b = c.type == 1 ? 2 : 4; cbytes.push_back(b); cwords.push_back(b / 2); line_bytes += (size_t) b * w; }
    int nblocks = (h + lines_per_block - 1) / lines_per_block;
    need(i, (size_t) nblocks * 8);
    Image img; img.width = w; img.height = h; img.channels = (int) chans.size(); img.bits_per_channel = 32; img.srgb = false;
    for (auto &c : chans) img.channel_names.push_back(c.name);
    img.data.assign((size_t) w * h * chans.size(), 0.f);
    for (int b = 0; b < nblocks; ++b) {
        uint64_t off; memcpy(&off, &f[i + 8 * (size_t) b], 8);
        need(off, 8); int32_t y0, dsz; memcpy(&y0, &f[off], 4); memcpy(&dsz, &f[off + 4], 4);
        need(off + 8, (size_t) dsz);
        const uint8_t *src = &f[off + 8];
        int ny = std::min(lines_per_block, dw[3] - y0 + 1);
        size_t raw_size = line_bytes * ny;
        std::vector<uint8_t> raw;
        if ((size_t) dsz >= raw_size) raw.assign(src, src + raw_size);
        else if (compression == 2 || compression == 3) {
            std::vector<uint8_t> t = zlib_inflate(src, (size_t) dsz, raw_size);
            if (t.size() != raw_size) throw std::runtime_error("\"" + path + "\": bad ZIP block size");
            for (size_t k = 1; k < t.size(); ++k) t[k] = (uint8_t) (t[k - 1] + t[k] - 128);
            raw.resize(raw_size); size_t half = (raw_size + 1) / 2;
            for (size_t k = 0; k < raw_size; ++k) raw[k] = (k & 1) ? t[half + k / 2] : t[k / 2];
        } else {
            std::vector<uint16_t> words; piz::decompress(src, dsz, w, ny, cwords, words);
            raw.resize(raw_size); memcpy(raw.data(), words.data(), raw_size);
        }
        const uint8_t *p = raw.data();
        for (int y = 0; y < ny; ++y) {
            int yy = y0 - dw[1] + y;
            for (size_t c = 0; c < chans.size(); ++c)
                for (int x = 0; x < w; ++x) {
                    float v;
                    if (chans[c].type == 1) { uint16_t hv; memcpy(&hv, p, 2); p += 2; v = half_to_float(hv); }
                    else if (chans[c].type == 2) { memcpy(&v, p, 4); p += 4; }
                    else { uint32_t u; memcpy(&u, p, 4); p += 4; v = (float) u; }
                    img.data[((size_t) yy * w + x) * chans.size() + c] = v;
                }
        }
    }
    return img;
}

Image read_image_rgb(const std::string &path) {
    std::string ext; size_t d = path.rfind('.'); if (d != std::string::npos) ext = path.substr(d + 1);
    for (auto &c : ext) c = (char) tolower(c);
    if (ext == "png") return read_png(path);
    if (ext == "exr") {
        Image e = read_exr(path);
        // reorder named channels to R,G,B[,A] / Y
        auto idx = [&](const char *n) { for (size_t k = 0; k < e.channel_names.size(); ++k) if (e.channel_names[k] == n) return (int) k; return -1; };
        int r = idx("R"), g = idx("G"), b = idx("B"), a = idx("A"), yc = idx("Y");
        Image o; o.width = e.width; o.height = e.height; o.bits_per_channel = 32; o.srgb = false;
        std::vector<int> map;
        if (r >= 0 && g >= 0 && b >= 0) { map = { r, g, b }; if (a >= 0) map.push_back(a); }
        else if (yc >= 0) { map = { yc }; if (a >= 0) map.push_back(a); }
        else throw std::runtime_error("\"" + path + "\": EXR has neither RGB nor Y channels");
        o.channels = (int) map.size(); o.data.resize((size_t) o.width * o.height * o.channels);
        for (size_t p = 0; p < (size_t) o.width * o.height; ++p)
            for (size_t c = 0; c < map.size(); ++c) o.data[p * map.size() + c] = e.data[p * e.channels + map[c]];
        return o;
    }
    throw std::runtime_error("\"" + path + "\": unsupported image format (only .png and .exr)");
}

void write_exr(const std::string &path, int w, int h, int channels, const float *data) {
    if (channels != 3 && channels != 4 && channels != 1) throw std::runtime_error("write_exr: 1, 3 or 4 channels expected");
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot open \"" + path + "\" for writing");
    auto put = [&](const void *p, size_t n) { fwrite(p, 1, n, f); };
    auto attr = [&](const char *name, const char *type, const void *v, uint32_t n) { put(name, strlen(name) + 1); put(type, strlen(type) + 1); put(&n, 4); put(v, n); };
    const uint8_t magic[8] = { 0x76, 0x2f, 0x31, 0x01, 2, 0, 0, 0 }; put(magic, 8);
    std::vector<std::string> names = channels == 1 ? std::vector<std::string>{ "Y" } : channels == 3 ? std::vector<std::string>{ "B", "G", "R" } : std::vector<std::string>{ "A", "B", "G", "R" };
    std::vector<uint8_t> cl;
    for (auto &n : names) { cl.insert(cl.end(), n.begin(), n.end()); cl.push_back(0); int32_t v[4] = { 2, 0, 1, 1 }; const uint8_t *q = (const uint8_t *) v; cl.insert(cl.end(), q, q + 16); }
    cl.push_back(0);
    attr("channels", "chlist", cl.data(), (uint32_t) cl.size());
    uint8_t comp = 0; attr("compression", "compression", &comp, 1);
    int32_t box[4] = { 0, 0, w - 1, h - 1 }; attr("dataWindow", "box2i", box, 16); attr("displayWindow", "box2i", box, 16);
    uint8_t lo = 0; attr("lineOrder", "lineOrder", &lo, 1);
    float one = 1.f, zero2[2] = { 0.f, 0.f }; attr("pixelAspectRatio", "float", &one, 4);
    attr("screenWindowCenter", "v2f", zero2, 8); attr("screenWindowWidth", "float", &one, 4);
    uint8_t z = 0; put(&z, 1);
    size_t line = (size_t) w * channels * 4; uint64_t off = (uint64_t) ftell(f) + 8ull * h;
    for (int y = 0; y < h; ++y) { put(&off, 8); off += 8 + line; }
    std::vector<float> row((size_t) w * channels);
    // file channel order is alphabetical: (A,)B,G,R  <- source order R,G,B(,A)
    std::vector<int> src = channels == 1 ? std::vector<int>{ 0 } : channels == 3 ? std::vector<int>{ 2, 1, 0 } : std::vector<int>{ 3, 2, 1, 0 };
    for (int y = 0; y < h; ++y) {
        int32_t hdr[2] = { y, (int32_t) line }; put(hdr, 8);
        for (int c = 0; c < channels; ++c) for (int x = 0; x < w; ++x) row[(size_t) c * w + x] = data[((size_t) y * w + x) * channels + src[c]];
        put(row.data(), line);
    }
    fclose(f);
}

// 8-bit PNG export of a linear float image, as LiverRenderer.py:383-385 does with
// Bitmap.convert(RGBA, UInt8, srgb_gamma=True): colour channels through the sRGB OETF, alpha linear, clamp to [0,1],
// round to nearest.  Filter type 0, one zlib stream, CRCs from zlib.
void write_png(const std::string &path, int w, int h, int channels, const float *data) {
    if (channels < 1 || channels > 4) throw std::runtime_error("write_png: 1 to 4 channels expected");
    const bool has_alpha = channels == 2 || channels == 4;
    const int n_colour = has_alpha ? channels - 1 : channels;
    std::vector<uint8_t> raw((size_t) h * ((size_t) w * channels + 1));
    size_t k = 0;
    for (int y = 0; y < h; ++y) {
        raw[k++] = 0;
        for (int x = 0; x < w; ++x) for (int c = 0; c < channels; ++c) {
            float v = data[((size_t) y * w + x) * channels + c];
            if (!(v > 0.f)) v = 0.f;                       // also NaN
            if (c < n_colour) v = v <= 0.0031308f ? 12.92f * v : 1.055f * std::pow(v, 1.f / 2.4f) - 0.055f;
            if (v > 1.f) v = 1.f;
            raw[k++] = (uint8_t) std::lround(v * 255.f);
        }
    }
    uLongf zn = compressBound((uLong) raw.size());
    std::vector<uint8_t> z(zn);
    if (compress2(z.data(), &zn, raw.data(), (uLong) raw.size(), 6) != Z_OK) throw std::runtime_error("write_png: deflate failed");
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot open \"" + path + "\" for writing");
    auto be32 = [](uint32_t v, uint8_t *p) { p[0] = (uint8_t) (v >> 24); p[1] = (uint8_t) (v >> 16); p[2] = (uint8_t) (v >> 8); p[3] = (uint8_t) v; };
    auto chunk = [&](const char *type, const uint8_t *p, uint32_t n) {
        uint8_t len[4]; be32(n, len); fwrite(len, 1, 4, f); fwrite(type, 1, 4, f); if (n) fwrite(p, 1, n, f);
        uLong crc = crc32(0L, (const Bytef *) type, 4); if (n) crc = crc32(crc, p, n);
        uint8_t c[4]; be32((uint32_t) crc, c); fwrite(c, 1, 4, f);
    };
    const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a }; fwrite(sig, 1, 8, f);
    uint8_t ihdr[13]; be32((uint32_t) w, ihdr); be32((uint32_t) h, ihdr + 4);
    ihdr[8] = 8; ihdr[9] = channels == 1 ? 0 : channels == 2 ? 4 : channels == 3 ? 2 : 6; ihdr[10] = ihdr[11] = ihdr[12] = 0;
    chunk("IHDR", ihdr, 13); chunk("IDAT", z.data(), (uint32_t) zn); chunk("IEND", nullptr, 0);
    fclose(f);
}

} // namespace lrt
