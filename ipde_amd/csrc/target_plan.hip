// Host-side cut of a target list into 4 x 4 tensor patches for ipde_laplace_apply_patches
// (no device code: plain C++ behind the C ABI, so that a cold process pays no GPU-library
// start-up for it; ipde_amd/target_plan.py holds the same algorithm in torch for lists that
// only exist on the device).
//
// The reference's big target lists are grid lists — grid_pnai: the points of a regular grid in C
// order outside the annuli, followed by the interface nodes (ipde/ebdy_collection.py:426-429).
// Steps: (1) distinct x and y VALUES with their multiplicities (exact comparisons; a value at
// least `line_min_points` points share is a grid line), (2) the lattice of the lines with the
// list position of the point at every crossing, (3) 4 x 4 tiles of the lattice: all tiles that
// hold a point if the list fills at least `partial_min_fill` of them (missing points become
// unstored ones, pout = -1), else the full tiles only, (4) tiles in the order lanes take them:
// blocks of block_i x block_j tiles, (5) everything else — off-lattice points, repeated points,
// points of unused tiles — in list order as the remainder.
#include "ipde_common.h"

#include <algorithm>
#include <cmath>
#include <new>

struct ipde_target_plan {
    int64_t n = 0, np = 0, nrest = 0;
    std::vector<double> pxy;     // 8 x np
    std::vector<int32_t> pout;   // 16 x np
    std::vector<int64_t> rest;   // nrest
};

namespace {

// distinct doubles (operator== semantics: -0.0 is 0.0, a NaN equals nothing) -> dense ids
struct ValueIds {
    std::vector<double> key;
    std::vector<int32_t> id;      // -1: empty
    std::vector<double> value;    // id -> value
    std::vector<int32_t> count;   // id -> multiplicity
    size_t mask = 0;
    explicit ValueIds(size_t cap_pow2) : key(cap_pow2), id(cap_pow2, -1), mask(cap_pow2 - 1) {}
    static size_t hash(double v) {
        uint64_t b;
        v += 0.0;                 // -0.0 -> +0.0
        memcpy(&b, &v, 8);
        b ^= b >> 29;
        b *= 0x9E3779B97F4A7C15ull;
        return (size_t)(b >> 20);
    }
    void grow() {
        std::vector<double> k2(key.size() * 2);
        std::vector<int32_t> i2(key.size() * 2, -1);
        const size_t m2 = k2.size() - 1;
        for (size_t s = 0; s < key.size(); ++s)
            if (id[s] >= 0) {
                size_t h = hash(key[s]) & m2;
                while (i2[h] >= 0) h = (h + 1) & m2;
                k2[h] = key[s];
                i2[h] = id[s];
            }
        key.swap(k2);
        id.swap(i2);
        mask = m2;
    }
    int32_t get(double v) {
        if (v != v) {             // NaN: a value of its own every time
            value.push_back(v);
            count.push_back(1);
            return (int32_t)value.size() - 1;
        }
        size_t h = hash(v) & mask;
        while (id[h] >= 0) {
            if (key[h] == v) {
                ++count[id[h]];
                return id[h];
            }
            h = (h + 1) & mask;
        }
        if (2 * (value.size() + 1) > key.size()) {
            grow();
            h = hash(v) & mask;
            while (id[h] >= 0) h = (h + 1) & mask;
        }
        key[h] = v;
        id[h] = (int32_t)value.size();
        value.push_back(v);
        count.push_back(1);
        return id[h];
    }
};

// ids of the values that are grid lines -> rank among the lines (sorted by value), others -1
int64_t rank_lines(const ValueIds& t, int line_min_points, std::vector<int32_t>& slot, std::vector<double>& lines) {
    std::vector<std::pair<double, int32_t>> keep;
    for (size_t i = 0; i < t.value.size(); ++i)
        if (t.count[i] >= line_min_points && std::isfinite(t.value[i])) keep.emplace_back(t.value[i], (int32_t)i);
    std::sort(keep.begin(), keep.end());
    slot.assign(t.value.size(), -1);
    lines.resize(keep.size());
    for (size_t r = 0; r < keep.size(); ++r) {
        slot[keep[r].second] = (int32_t)r;
        lines[r] = keep[r].first;
    }
    return (int64_t)keep.size();
}

void all_rest(ipde_target_plan& p) {
    p.np = 0;
    p.pxy.clear();
    p.pout.clear();
    p.nrest = p.n;
    p.rest.resize(p.n);
    for (int64_t i = 0; i < p.n; ++i) p.rest[i] = i;
}

}  // namespace

extern "C" int ipde_target_plan_build_blocks(int64_t nt, const double* x, const double* y, int block_i, int block_j,
                                             double partial_min_fill, int64_t min_patches, int line_min_points,
                                             int pad_blocks, ipde_target_plan** out);

extern "C" int ipde_target_plan_build(int64_t nt, const double* x, const double* y, int block_i, int block_j,
                                      double partial_min_fill, int64_t min_patches, int line_min_points,
                                      ipde_target_plan** out) {
    return ipde_target_plan_build_blocks(nt, x, y, block_i, block_j, partial_min_fill, min_patches,
                                         line_min_points, 0, out);
}

// pad_blocks != 0: every block of block_i x block_j tiles that holds a patch is filled up to
// block_i * block_j patches with copies of its first one that store nothing (pout = -1), so that
// patches [k B, (k + 1) B), B = block_i block_j, are exactly one block — what
// ipde_laplace_apply_patches_far takes (B = 64: a wave per block).
extern "C" int ipde_target_plan_build_blocks(int64_t nt, const double* x, const double* y, int block_i, int block_j,
                                             double partial_min_fill, int64_t min_patches, int line_min_points,
                                             int pad_blocks, ipde_target_plan** out) {
    if (!out) return IPDE_ERR_INVALID;
    *out = nullptr;
    if (nt < 0 || nt >= (1LL << 31) || (nt > 0 && (!x || !y)) || block_i < 1 || block_j < 1 || line_min_points < 1)
        return IPDE_ERR_INVALID;
    try {
        ipde_target_plan* p = new ipde_target_plan();
        *out = p;
        p->n = nt;
        if (nt < 16) {
            all_rest(*p);
            return IPDE_OK;
        }
        // (1) distinct values.  x of a C-ordered grid list is constant over long runs: one look-up per run
        std::vector<int32_t> idx(nt), idy(nt);
        ValueIds tx(1 << 14), ty(1 << 14);
        for (int64_t i = 0; i < nt; ++i) {
            if (i > 0 && x[i] == x[i - 1]) {
                idx[i] = idx[i - 1];
                ++tx.count[idx[i]];
            } else {
                idx[i] = tx.get(x[i]);
            }
            idy[i] = ty.get(y[i]);
        }
        std::vector<int32_t> slotx, sloty;
        std::vector<double> ux, uy;
        const int64_t nx = rank_lines(tx, line_min_points, slotx, ux);
        const int64_t ny = rank_lines(ty, line_min_points, sloty, uy);
        if (nx < 4 || ny < 4 || nx * ny > 4 * nt + 4096) {   // scattered points: a lattice of holes
            all_rest(*p);
            return IPDE_OK;
        }
        // (2) the lattice
        const int64_t nxp = (nx + 3) / 4 * 4, nyp = (ny + 3) / 4 * 4, ni = nxp / 4, nj = nyp / 4;
        std::vector<int32_t> pos((size_t)nxp * nyp, -1);
        std::vector<uint8_t> owner(nt, 0);
        for (int64_t i = 0; i < nt; ++i) {
            const int32_t a = slotx[idx[i]], b = sloty[idy[i]];
            if (a < 0 || b < 0) continue;
            int32_t& slot = pos[(size_t)a * nyp + b];
            if (slot < 0) {          // a repeated point: the first one sits on the lattice
                slot = (int32_t)i;
                owner[i] = 1;
            }
        }
        // (3) tiles
        std::vector<uint8_t> cnt((size_t)ni * nj, 0);
        int64_t present = 0, nonempty = 0;
        for (int64_t I = 0; I < ni; ++I)
            for (int64_t J = 0; J < nj; ++J) {
                int c = 0;
                for (int a = 0; a < 4; ++a)
                    for (int b = 0; b < 4; ++b) c += pos[(size_t)(4 * I + a) * nyp + 4 * J + b] >= 0;
                cnt[(size_t)I * nj + J] = (uint8_t)c;
                present += c;
                nonempty += c > 0;
            }
        const bool partial = (double)present >= partial_min_fill * 16.0 * (double)nonempty;
        auto used = [&](int64_t I, int64_t J) {
            const int c = cnt[(size_t)I * nj + J];
            return partial ? c > 0 : c == 16;
        };
        int64_t np = 0;
        for (int64_t I = 0; I < ni; ++I)
            for (int64_t J = 0; J < nj; ++J) np += used(I, J);
        if (np < std::max<int64_t>(1, min_patches)) {
            all_rest(*p);
            return IPDE_OK;
        }
        // (4) patches in the lanes' order
        if (pad_blocks) {
            const int64_t B = (int64_t)block_i * block_j;
            int64_t nblocks = 0;
            for (int64_t Ib = 0; Ib < ni; Ib += block_i)
                for (int64_t Jb = 0; Jb < nj; Jb += block_j) {
                    bool any = false;
                    for (int64_t I = Ib; I < std::min<int64_t>(ni, Ib + block_i) && !any; ++I)
                        for (int64_t J = Jb; J < std::min<int64_t>(nj, Jb + block_j) && !any; ++J) any = used(I, J);
                    nblocks += any;
                }
            np = nblocks * B;
            if (np >= (1LL << 27)) {
                all_rest(*p);
                return IPDE_OK;
            }
        }
        p->np = np;
        p->pxy.resize((size_t)8 * np);
        p->pout.resize((size_t)16 * np);
        // blocks in row-major order of their tile coordinates; padded plans: in Z (Morton) order, so
        // that 16 consecutive blocks are a 4 x 4 group of blocks, 256 a 16 x 16 one (the far-field
        // kernels' second level takes 16 consecutive blocks as one parent block)
        std::vector<std::pair<uint64_t, std::pair<int64_t, int64_t>>> blocks;
        for (int64_t Ib = 0; Ib < ni; Ib += block_i)
            for (int64_t Jb = 0; Jb < nj; Jb += block_j) {
                uint64_t key = (uint64_t)blocks.size();
                if (pad_blocks) {
                    const uint64_t a = (uint64_t)(Ib / block_i), b = (uint64_t)(Jb / block_j);
                    key = 0;
                    for (int bit = 0; bit < 32; ++bit)
                        key |= ((a >> bit) & 1ull) << (2 * bit + 1) | ((b >> bit) & 1ull) << (2 * bit);
                }
                blocks.push_back({key, {Ib, Jb}});
            }
        if (pad_blocks) std::sort(blocks.begin(), blocks.end());
        int64_t q = 0;
        for (const auto& blk : blocks) {
            const int64_t Ib = blk.second.first, Jb = blk.second.second;
            const int64_t q0 = q;
            for (int64_t I = Ib; I < std::min<int64_t>(ni, Ib + block_i); ++I)
                for (int64_t J = Jb; J < std::min<int64_t>(nj, Jb + block_j); ++J) {
                    if (!used(I, J)) continue;
                    for (int a = 0; a < 4; ++a) {
                        p->pxy[(size_t)a * np + q] = ux[std::min<int64_t>(nx - 1, 4 * I + a)];
                        p->pxy[(size_t)(4 + a) * np + q] = uy[std::min<int64_t>(ny - 1, 4 * J + a)];
                        for (int b = 0; b < 4; ++b)
                            p->pout[(size_t)(4 * a + b) * np + q] = pos[(size_t)(4 * I + a) * nyp + 4 * J + b];
                    }
                    ++q;
                }
            if (pad_blocks && q > q0)
                for (const int64_t qe = q0 + (int64_t)block_i * block_j; q < qe; ++q) {
                    for (int r = 0; r < 8; ++r) p->pxy[(size_t)r * np + q] = p->pxy[(size_t)r * np + q0];
                    for (int r = 0; r < 16; ++r) p->pout[(size_t)r * np + q] = -1;
                }
        }
        // (5) the remainder, in list order
        for (int64_t i = 0; i < nt; ++i) {
            bool in_patch = false;
            if (owner[i]) {
                const int32_t a = slotx[idx[i]], b = sloty[idy[i]];
                in_patch = used(a / 4, b / 4);
            }
            if (!in_patch) p->rest.push_back(i);
        }
        p->nrest = (int64_t)p->rest.size();
        return IPDE_OK;
    } catch (const std::bad_alloc&) {
        delete *out;
        *out = nullptr;
        return IPDE_ERR_ALLOC;
    }
}

extern "C" int ipde_target_plan_sizes(const ipde_target_plan* p, int64_t* np, int64_t* nrest) {
    if (!p || !np || !nrest) return IPDE_ERR_INVALID;
    *np = p->np;
    *nrest = p->nrest;
    return IPDE_OK;
}

extern "C" int ipde_target_plan_export(const ipde_target_plan* p, double* pxy, int32_t* pout, int64_t* rest) {
    if (!p) return IPDE_ERR_INVALID;
    if ((p->np > 0 && (!pxy || !pout)) || (p->nrest > 0 && !rest)) return IPDE_ERR_INVALID;
    if (p->np > 0) {
        memcpy(pxy, p->pxy.data(), p->pxy.size() * sizeof(double));
        memcpy(pout, p->pout.data(), p->pout.size() * sizeof(int32_t));
    }
    if (p->nrest > 0) memcpy(rest, p->rest.data(), p->rest.size() * sizeof(int64_t));
    return IPDE_OK;
}

extern "C" int ipde_target_plan_destroy(ipde_target_plan* p) {
    delete p;
    return IPDE_OK;
}
