// Pair-grid tiling and scheduling (host code, no device work): the N x N comparison grid of
// CoverAlgorithm.all_pairwise (reference acoss/algorithms/algorithm_template.py:168-191: the pair
// list from itertools.combinations / permutations, 45 joblib chunks, D += D.T) cut into B x B track
// tiles, costed by sum len_i len_j, sorted and dealt to the ranks longest-processing-time-first.
// The plan is a pure function of (track lengths, spec): every rank computes the same one.
#pragma once
#include <algorithm>
#include <cstdint>
#include <queue>
#include <vector>

#include "../../include/acx.h"

namespace acx {

inline int grid_planes(int algo)
{
    switch (algo) {
    case ACX_ALGO_SERRA09: return 1;
    case ACX_ALGO_CHENFUSION: return 2;
    case ACX_ALGO_SIMPLE: return 1;
    case ACX_ALGO_EARLYFUSION: return 4;
    default: return 0;
    }
}

inline bool grid_spec_ok(const acx_grid_spec *s)
{
    return s && grid_planes(s->algo) > 0 && s->world >= 1 && s->tile >= 0 && (s->symmetric == 0 || s->symmetric == 1);
}

// tile edge: the caller's, or 128 tracks shrunk until every rank has a few dozen tiles to balance with
inline int grid_tile_edge(const acx_grid_spec &s, int n)
{
    if (s.tile > 0) return s.tile;
    int t = 128;
    auto ntiles = [&](int e) {
        const int64_t nb = (n + e - 1) / e;
        return s.symmetric ? nb * (nb + 1) / 2 : nb * nb;
    };
    while (t > 8 && ntiles(t) < (int64_t)32 * s.world) t /= 2;
    return t;
}

// Tiles in deal order (cost descending; ties by (row0, col0)), each with its owner and the offset of
// its rows x cols x planes scores in the owner's buffer.
inline void grid_plan(const int64_t *len, int n, const acx_grid_spec &s, std::vector<acx_grid_tile> &tiles,
                      std::vector<int64_t> &floats_per_rank, std::vector<double> &cost_per_rank)
{
    const int e = grid_tile_edge(s, n), w = grid_planes(s.algo);
    const int nb = (n + e - 1) / e;
    std::vector<double> S(nb, 0.0), Q(nb, 0.0);
    for (int i = 0; i < n; ++i) {
        const double l = (double)len[i];
        S[i / e] += l;
        Q[i / e] += l * l;
    }
    tiles.clear();
    for (int a = 0; a < nb; ++a)
        for (int b = s.symmetric ? a : 0; b < nb; ++b) {
            acx_grid_tile t;
            t.row0 = a * e; t.col0 = b * e;
            t.rows = std::min(e, n - t.row0); t.cols = std::min(e, n - t.col0);
            t.diagonal = (a == b) ? 1 : 0;
            t.rank = 0; t.offset = 0;
            if (a != b) t.cost = S[a] * S[b];
            else t.cost = s.symmetric ? 0.5 * (S[a] * S[a] - Q[a]) : (S[a] * S[a] - Q[a]);
            tiles.push_back(t);
        }
    std::stable_sort(tiles.begin(), tiles.end(), [](const acx_grid_tile &x, const acx_grid_tile &y) { return x.cost > y.cost; });
    floats_per_rank.assign(s.world, 0);
    cost_per_rank.assign(s.world, 0.0);
    typedef std::pair<double, int> Load;      // (load, rank): least loaded first, lowest rank on ties
    std::priority_queue<Load, std::vector<Load>, std::greater<Load>> heap;
    for (int r = 0; r < s.world; ++r) heap.push(Load(0.0, r));
    for (acx_grid_tile &t : tiles) {
        const Load top = heap.top();
        heap.pop();
        t.rank = top.second;
        t.offset = floats_per_rank[t.rank];
        floats_per_rank[t.rank] += (int64_t)t.rows * t.cols * w;
        cost_per_rank[t.rank] += t.cost;
        heap.push(Load(top.first + t.cost, t.rank));
    }
}

// this rank's tiles [first, first + count) in deal order (count < 0: to the end)
inline std::vector<acx_grid_tile> grid_slice(const std::vector<acx_grid_tile> &tiles, int rank, int64_t first, int64_t count)
{
    std::vector<acx_grid_tile> out;
    int64_t k = 0;
    for (const acx_grid_tile &t : tiles) {
        if (t.rank != rank) continue;
        if (k >= first && (count < 0 || k < first + count)) out.push_back(t);
        ++k;
    }
    return out;
}

// pairs of a tile in row-major order; a symmetric grid's diagonal tile holds i < j only, an ordered
// grid's i != j.  `idx`: float offset of the pair's first plane in the owner's buffer.
inline void grid_tile_pairs(const acx_grid_tile &t, int symmetric, int w, std::vector<int32_t> &pairs, std::vector<int64_t> &idx)
{
    for (int a = 0; a < t.rows; ++a)
        for (int b = 0; b < t.cols; ++b) {
            const int i = t.row0 + a, j = t.col0 + b;
            if (t.diagonal && (symmetric ? !(i < j) : i == j)) continue;
            pairs.push_back(i);
            pairs.push_back(j);
            idx.push_back(t.offset + ((int64_t)a * t.cols + b) * w);
        }
}

// Tile scores (gathered: `world` rank buffers of `rank_stride` floats each, host memory) into the
// N x N planes.  mirror: also D[j][i] = D[i][j] for every computed pair (the reference's D += D.T
// on a matrix whose lower triangle is still zero).  The transposed tile is written row by row.
inline void grid_scatter(const std::vector<acx_grid_tile> &tiles, const acx_grid_spec &s, const float *gathered,
                         int64_t rank_stride, int64_t first, int64_t count, float *const *D, int64_t ld, int mirror)
{
    const int w = grid_planes(s.algo);
    std::vector<int64_t> seen(s.world, 0);
    for (const acx_grid_tile &t : tiles) {
        const int64_t k = seen[t.rank]++;
        if (k < first || (count >= 0 && k >= first + count)) continue;
        const float *src = gathered + (int64_t)t.rank * rank_stride + t.offset;
        for (int e = 0; e < w; ++e) {
            float *P = D[e];
            for (int a = 0; a < t.rows; ++a) {
                float *row = P + (int64_t)(t.row0 + a) * ld + t.col0;
                for (int b = 0; b < t.cols; ++b) {
                    const int i = t.row0 + a, j = t.col0 + b;
                    if (t.diagonal && (s.symmetric ? !(i < j) : i == j)) continue;
                    row[b] = src[((int64_t)a * t.cols + b) * w + e];
                }
            }
            if (mirror) {
                for (int b = 0; b < t.cols; ++b) {
                    float *row = P + (int64_t)(t.col0 + b) * ld + t.row0;
                    for (int a = 0; a < t.rows; ++a) {
                        const int i = t.row0 + a, j = t.col0 + b;
                        if (t.diagonal && (s.symmetric ? !(i < j) : i == j)) continue;
                        row[a] = src[((int64_t)a * t.cols + b) * w + e];
                    }
                }
            }
        }
    }
}

}  // namespace acx
