// rle.hip -- host-side run-length codec for CARLE universes.  HOST pointers.
//
// Restates the reference's get_rle body (carle/env.py:431-462) and rle_to_grid (carle/env.py:260-328)
// as linear-time C++ over byte cells, replacing per-cell Python/tensor indexing.  The encoder is
// byte-compatible with the reference, including its quirks: run counts are always written ("1o"),
// a line is flushed once it is longer than 69 characters, and whatever is still buffered when the
// last row ends is DROPPED before the closing "!" (the reference never appends it, env.py:459-462).
#include <stdint.h>
#include <string>

#include "common.hpp"

using namespace carle;

extern "C" int64_t carle_rle_encode_body(const uint8_t *cells, int h, int w, char *out, int64_t cap) {
    if (!cells || h < 1 || w < 1) return -1;
    std::string rle, line;
    line.reserve(96);
    for (int y = 0; y < h; ++y) {
        const uint8_t *row = cells + (int64_t)y * w;
        uint8_t cur = row[0] != 0;
        int run = 1;
        for (int x = 1; x < w; ++x) {
            const uint8_t v = row[x] != 0;
            if (v == cur) {
                ++run;
            } else {
                line += std::to_string(run);
                line += cur ? 'o' : 'b';
                if (line.size() > 69) { rle += line; rle += '\n'; line.clear(); }
                cur = v;
                run = 1;
            }
        }
        line += std::to_string(run);
        line += cur ? 'o' : 'b';
        line += '$';
        if (line.size() > 69) { rle += line; rle += '\n'; line.clear(); }
    }
    rle += '!';                                   // buffered tail intentionally not appended
    const int64_t need = (int64_t)rle.size();
    if (out && cap > 0) {
        const int64_t k = need < cap ? need : cap;
        for (int64_t i = 0; i < k; ++i) out[i] = rle[(size_t)i];
    }
    return need;
}

// returns 0 on success, CARLE_E_DIMS if a run starts on a row outside the grid (the reference
// raises IndexError there), CARLE_E_FORMAT on a malformed count
extern "C" int carle_rle_decode_body(const char *rle, int64_t len, uint8_t *cells, int h, int w) {
    if (!rle || !cells || h < 1 || w < 1 || len < 0) return fail_arg(CARLE_E_NULL, "carle_rle_decode_body: bad argument");
    for (int64_t i = 0; i < (int64_t)h * w; ++i) cells[i] = 0;
    int64_t ii = 0, jj = 0;
    std::string temp;
    auto count_of = [&](bool &ok) -> int64_t {           // int(temp[:-1]); "b"/"o"/"$" alone -> 1
        ok = true;
        if (temp.size() == 1) return 1;
        int64_t v = 0;
        for (size_t k = 0; k + 1 < temp.size(); ++k) {
            const char c = temp[k];
            if (c == ' ') continue;
            if (c < '0' || c > '9') { ok = false; return 0; }
            v = v * 10 + (c - '0');
            if (v > (1ll << 40)) { ok = false; return 0; }
        }
        return v;
    };
    for (int64_t p = 0; p < len; ++p) {
        const char ch = rle[p];
        if (ch == '\n' || ch == '\r') continue;
        temp += ch;
        const char lo = (ch >= 'A' && ch <= 'Z') ? (char)(ch + 32) : ch;
        if (lo == 'b' || lo == 'o') {
            bool ok;
            const int64_t run = count_of(ok);
            if (!ok) return fail_arg(CARLE_E_FORMAT, "carle_rle_decode_body: malformed run count");
            if (ii < 0 || ii >= h) return fail_arg(CARLE_E_DIMS, "carle_rle_decode_body: pattern taller than the grid");
            const int64_t x1 = jj + run < w ? jj + run : w;
            for (int64_t x = jj; x < x1; ++x) cells[ii * w + x] = (lo == 'o');
            jj += run;
            temp.clear();
        } else if (ch == '$') {
            bool ok;
            const int64_t rows = count_of(ok);
            if (!ok) return fail_arg(CARLE_E_FORMAT, "carle_rle_decode_body: malformed row count");
            if (temp.size() > 1) {
                for (int64_t y = ii + 1; y < ii + rows && y < h; ++y)
                    for (int64_t x = 0; x < w; ++x) cells[y * w + x] = 0;
                ii += rows;
            } else {
                if (ii >= 0 && ii < h)
                    for (int64_t x = jj; x < w; ++x) cells[ii * w + x] = 0;
                ii += 1;
            }
            jj = 0;
            temp.clear();
        } else if (ch == '!') {
            temp.clear();
        }
    }
    return 0;
}
