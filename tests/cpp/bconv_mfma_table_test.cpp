// CPU check of the matrix-core base conversion's arithmetic (numth.hpp: build_bconv_mfma_table; rns_kernels.cuh:
// bconv_mfma_kernel): the table, walked exactly as the kernel walks it with v_mfma_i32_16x16x64_i8 emulated as
//   D[row][col] = sum over lane groups g' and bytes jb of  A(lane 16 g' + row).byte[jb] * B(lane 16 g' + col).byte[jb],
// must give sum_i y_i M[j][i] mod p_j for every output.  Needs no GPU.  Usage: bconv_mfma_table_test  (exit 0 = pass)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../phantom-fhe_amd/csrc/numth.hpp"

using namespace pha;
typedef unsigned __int128 u128;

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}

static uint64_t finish(const int32_t d0[4], const int32_t d1[4], uint64_t p, uint64_t mu) {
    // the kernel's bm_finish; registers 0 and 2 of each accumulator start at kBmfBiasWord
    const uint32_t b = kBmfBiasWord;
    const uint32_t lo0 = static_cast<uint32_t>(d0[0]) + b + (static_cast<uint32_t>(d0[1]) << 8);
    const uint32_t hi0 = static_cast<uint32_t>(d0[2]) + b + (static_cast<uint32_t>(d0[3]) << 8);
    const uint32_t lo1 = static_cast<uint32_t>(d1[0]) + b + (static_cast<uint32_t>(d1[1]) << 8);
    const uint32_t hi1 = static_cast<uint32_t>(d1[2]) + b + (static_cast<uint32_t>(d1[3]) << 8);
    for (uint32_t w : {lo0, hi0, lo1, hi1})
        if (w <= (1u << 29) || w >= (1u << 31)) {
            printf("accumulator word out of range\n");
            exit(1);
        }
    const uint64_t v0 = static_cast<uint64_t>(hi0) * 65536 + lo0, v1 = static_cast<uint64_t>(hi1) * 65536 + lo1;
    const uint64_t xh = v1 + (v0 >> 32), xlo = (xh << 32) | static_cast<uint32_t>(v0);
    const uint64_t qhat = static_cast<uint64_t>((static_cast<u128>(xh) * mu) >> 64);
    uint64_t r = xlo - qhat * p;
    if (r >= 2 * p) {
        printf("remainder estimate out of range\n");
        exit(1);
    }
    if (r >= p) r -= p;
    return r;
}

static int run_case(size_t ibase, size_t obase, int bits_lo, int bits_hi) {
    std::vector<uint64_t> mod(obase), matrix(obase * ibase);
    for (size_t j = 0; j < obase; j++) {
        const int bits = bits_lo + static_cast<int>(rnd() % (bits_hi - bits_lo + 1));
        mod[j] = ((1ull << (bits - 1)) | (rnd() & ((1ull << (bits - 1)) - 1))) | 1;   // any odd modulus of that size will do
        for (size_t i = 0; i < ibase; i++) matrix[j * ibase + i] = rnd() % mod[j];
    }
    matrix[0] = mod[0] - 1;
    BconvMfmaTable t;
    build_bconv_mfma_table(matrix.data(), mod.data(), ibase, obase, t);
    if (!t.ksteps) {
        printf("no table for ibase %zu\n", ibase);
        return 1;
    }
    const uint32_t ks_n = t.ksteps;
    const size_t quad_bytes = 2 * ks_n * 1024;
    // 16 coefficients (one MFMA column group), extremes included
    uint64_t y[16][16];
    for (size_t i = 0; i < 16; i++)
        for (int c = 0; c < 16; c++) y[i][c] = c == 0 ? ~0ull : c == 1 ? 0 : c == 2 ? 0x8080808080808080ull : c == 3 ? 0x7F7F7F7F7F7F7F7Full : rnd();
    // B operands: lane (g, c), k-step ks: bytes of inputs 8 ks + 2 g + h, biased; the constant slot
    auto b_byte = [&](uint32_t ks, uint32_t g, uint32_t c, uint32_t jb) -> int {
        const size_t i = ks * 8 + 2 * g + jb / 8, a = jb % 8;
        if (i < ibase) return static_cast<int8_t>(static_cast<uint8_t>(y[i][c] >> (8 * a)) ^ 0x80);
        if (i == ibase && a == 0) return 1;
        return 0;
    };
    int bad = 0;
    for (size_t quad = 0; quad < t.quads; quad++)
        for (uint32_t g = 0; g < 4; g++)
            for (uint32_t c = 0; c < 16; c++) {
                const size_t j = quad * 4 + g;
                if (j >= obase) continue;
                int32_t d[2][4];
                for (uint32_t tile = 0; tile < 2; tile++)
                    for (uint32_t r = 0; r < 4; r++) {
                        int64_t acc = 0;
                        for (uint32_t ks = 0; ks < ks_n; ks++)
                            for (uint32_t gk = 0; gk < 4; gk++)
                                for (uint32_t jb = 0; jb < 16; jb++) {
                                    const uint32_t lane = 16 * gk + 4 * g + r;
                                    const int8_t a = static_cast<int8_t>(t.table[quad * quad_bytes + ((tile * ks_n + ks) * 64 + lane) * 16 + jb]);
                                    acc += static_cast<int64_t>(a) * b_byte(ks, gk, c, jb);
                                }
                        if (acc >= (1 << 21) || acc <= -(1 << 21)) {
                            printf("partial sum out of range\n");
                            return 1;
                        }
                        d[tile][r] = static_cast<int32_t>(acc);
                    }
                const uint64_t got = finish(d[0], d[1], t.consts[j * 2], t.consts[j * 2 + 1]);
                u128 want = 0;
                for (size_t i = 0; i < ibase; i++) want = (want + static_cast<u128>(y[i][c]) * matrix[j * ibase + i]) % mod[j];
                if (got != static_cast<uint64_t>(want)) {
                    if (bad++ < 5) printf("ibase %zu obase %zu j %zu c %u: got %llu want %llu\n", ibase, obase, j, c, (unsigned long long)got, (unsigned long long)want);
                }
            }
    return bad;
}

int main() {
    int bad = 0;
    const size_t shapes[][2] = {{1, 5}, {2, 7}, {3, 4}, {7, 9}, {8, 13}, {12, 36}, {13, 35}, {15, 6}};
    for (const auto &s : shapes) {
        bad += run_case(s[0], s[1], 35, 61);
        bad += run_case(s[0], s[1], 61, 61);
        bad += run_case(s[0], s[1], 35, 36);
    }
    BconvMfmaTable t;
    std::vector<uint64_t> m(16 * 2, 1), p = {(1ull << 40) + 15, (1ull << 34) - 41};
    build_bconv_mfma_table(m.data(), p.data(), 16, 2, t);
    if (t.ksteps) bad++, printf("ibase 16 must have no table\n");
    build_bconv_mfma_table(m.data(), p.data(), 4, 2, t);
    if (t.ksteps) bad++, printf("a modulus below 2^34 must have no table\n");
    printf(bad ? "FAILED (%d)\n" : "ok\n", bad);
    return bad ? 1 : 0;
}
