// bvh_eval.cpp — offline quality figures of the 16-wide box-list hierarchy (rtx_bvh.cpp) for the cooperative shadow walk:
// builds it over a dumped triangle list and walks it with simulated shadow groups (16 feelers from one surface point to a
// light's sampling cube), counting what k_shadow_coop counts: node visits per group, own-box tests and exact tests per feeler.
// No GPU.  usage: bvh_eval scene.bin [groups]     (scene.bin: tools/exp/dump_scene.py)
// build: g++ -O2 -std=c++17 -pthread -I rust-raytracer_amd/csrc -I include -o /tmp/bvh/bvh_eval tools/exp/bvh_eval.cpp
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include <chrono>

#include "../../rust-raytracer_amd/csrc/rtx_bvh.cpp"

struct V { double x, y, z; };
static V sub(V a, V b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static V cross(V a, V b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static double dot(V a, V b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

struct Seg { V o, d, inv; double tmax; bool alive; };

static double g_tn;   // entry distance of the last seg_box() that passed
static bool seg_box(const Seg &s, const float *lo, const float *hi, const double *c) {
    double t0 = 0.0, t1 = s.tmax;
    const double o[3] = {s.o.x - c[0], s.o.y - c[1], s.o.z - c[2]}, inv[3] = {s.inv.x, s.inv.y, s.inv.z};
    for (int a = 0; a < 3; ++a) {
        double ta = (lo[a] - o[a]) * inv[a], tb = (hi[a] - o[a]) * inv[a];
        if (ta > tb) std::swap(ta, tb);
        if (ta > t0) t0 = ta;
        if (tb < t1) t1 = tb;
    }
    g_tn = t0;
    return t0 <= t1;
}
static bool seg_tri(const Seg &s, const double *v) {
    V v0{v[0], v[1], v[2]}, v1{v[3], v[4], v[5]}, v2{v[6], v[7], v[8]};
    V e1 = sub(v1, v0), e2 = sub(v2, v0), p = cross(s.d, e2);
    double det = dot(e1, p);
    if (det > -2.2e-16 && det < 2.2e-16) return false;
    double id = 1.0 / det;
    V sv = sub(s.o, v0);
    double b = id * dot(sv, p);
    if (b < 0.0 || b > 1.0) return false;
    V q = cross(sv, e1);
    double g = id * dot(s.d, q);
    if (g < 0.0 || b + g > 1.0) return false;
    double t = id * dot(e2, q);
    return t >= 2.2e-12 && t <= s.tmax;
}

int main(int argc, char **argv) {
    if (argc < 2) return 1;
    FILE *f = std::fopen(argv[1], "rb");
    int64_t hdr[2];
    if (!f || std::fread(hdr, 8, 2, f) != 2) return 1;
    const int64_t n = hdr[0], nl = hdr[1];
    std::vector<double> lights((size_t)nl * 4), tv((size_t)n * 9);
    if (std::fread(lights.data(), 8, lights.size(), f) != lights.size() || std::fread(tv.data(), 8, tv.size(), f) != tv.size()) return 1;
    std::fclose(f);
    const int groups = argc > 2 ? std::atoi(argv[2]) : 20000;

    std::vector<RtxPrimBox> prims((size_t)n);
    double smn[3] = {1e300, 1e300, 1e300}, smx[3] = {-1e300, -1e300, -1e300};
    std::vector<double> cum((size_t)n);
    double total_area = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        const double *v = &tv[(size_t)i * 9];
        RtxPrimBox &p = prims[(size_t)i];
        for (int a = 0; a < 3; ++a) {
            p.mn[a] = std::min(std::min(v[a], v[3 + a]), v[6 + a]);
            p.mx[a] = std::max(std::max(v[a], v[3 + a]), v[6 + a]);
            smn[a] = std::min(smn[a], p.mn[a]);
            smx[a] = std::max(smx[a], p.mx[a]);
        }
        p.ref = RTX_WREF_TRI | (uint32_t)i;
        V e1{v[3] - v[0], v[4] - v[1], v[5] - v[2]}, e2{v[6] - v[0], v[7] - v[1], v[8] - v[2]};
        V c = cross(e1, e2);
        total_area += 0.5 * std::sqrt(dot(c, c));
        cum[(size_t)i] = total_area;
    }
    double centre[3] = {0.5 * (smn[0] + smx[0]), 0.5 * (smn[1] + smx[1]), 0.5 * (smn[2] + smx[2])};
    RtxPool pool(std::getenv("BVH_THREADS") ? (unsigned)std::atoi(std::getenv("BVH_THREADS")) : 8u);
    std::vector<DWideChild> wide;
    RtxWideStats st;
    double ms = 1e30;
    for (int rep = 0; rep < (std::getenv("BVH_REPS") ? std::atoi(std::getenv("BVH_REPS")) : 1); ++rep) {   // best of BVH_REPS builds
        auto t0 = std::chrono::steady_clock::now();
        { std::vector<DNode4> b4; rtx_build_hierarchies(prims, centre, wide, b4, &st, pool); }
        ms = std::min(ms, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    const size_t n_nodes = wide.size() / RTX_WIDE;
    int hist[RTX_WIDE + 1] = {0};
    int64_t leafish = 0;
    for (size_t i = 0; i < n_nodes; ++i) {
        int c = 0, nn = 0;
        for (int s = 0; s < RTX_WIDE; ++s) {
            uint32_t k = wide[i * RTX_WIDE + s].ref & RTX_WREF_KIND;
            if (k != RTX_WREF_EMPTY) ++c;
            if (k == RTX_WREF_NODE) ++nn;
        }
        ++hist[c];
        if (nn == 0) ++leafish;
    }
    std::printf("build %.1f ms: %zu nodes (%lld without node children), depth %d, fill %.2f of 16\n  fill histogram:", ms, n_nodes,
                (long long)leafish, st.depth, (double)(st.prim_children + (int64_t)n_nodes - 1) / (double)n_nodes);
    for (int c = 0; c <= RTX_WIDE; ++c) std::printf(" %d", hist[c]);
    std::printf("\n");

    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    for (int64_t li = 0; li < nl; ++li) {
        const double *L = &lights[(size_t)li * 4];
        double by_depth[16] = {0}, by_fill[17] = {0};
        double s_nodes = 0, s_box = 0, s_exact = 0, s_queued = 0, s_occl = 0, s_steps = 0;
        for (int g = 0; g < groups; ++g) {
            // a surface point, area-weighted
            double r = U(rng) * total_area;
            size_t ti = (size_t)(std::lower_bound(cum.begin(), cum.end(), r) - cum.begin());
            if (ti >= (size_t)n) ti = (size_t)n - 1;
            const double *v = &tv[ti * 9];
            double a = U(rng), b = U(rng);
            if (a + b > 1.0) { a = 1.0 - a; b = 1.0 - b; }
            V o{v[0] + a * (v[3] - v[0]) + b * (v[6] - v[0]), v[1] + a * (v[4] - v[1]) + b * (v[7] - v[1]), v[2] + a * (v[5] - v[2]) + b * (v[8] - v[2])};
            Seg seg[16];
            for (int k = 0; k < 16; ++k) {
                V t{L[0] + L[3] * (U(rng) - 0.5), L[1] + L[3] * (U(rng) - 0.5), L[2] + L[3] * (U(rng) - 0.5)};
                V d = sub(t, o);
                double len = std::sqrt(dot(d, d));
                seg[k].o = o;
                seg[k].d = {d.x / len, d.y / len, d.z / len};
                seg[k].inv = {1.0 / seg[k].d.x, 1.0 / seg[k].d.y, 1.0 / seg[k].d.z};
                seg[k].tmax = len;
                seg[k].alive = true;
            }
            // depth-first walk, children in slot order; "beam touches box" = some live feeler's segment touches it
            std::vector<uint32_t> stack{0u};
            std::vector<int> dstack{0};
            int live = 16, visits = 0;
            while (!stack.empty() && live > 0) {
                uint32_t nd = stack.back();
                stack.pop_back();
                int dep = dstack.back();
                dstack.pop_back();
                by_depth[dep < 15 ? dep : 15] += 1;
                { int fc = 0; for (int s = 0; s < RTX_WIDE; ++s) fc += (wide[(size_t)nd * RTX_WIDE + s].ref & RTX_WREF_KIND) != RTX_WREF_EMPTY; by_fill[fc] += 1; }
                ++visits;
                std::vector<uint32_t> kids;
                std::vector<double> kid_tn;
                for (int s = 0; s < RTX_WIDE; ++s) {
                    const DWideChild &c = wide[(size_t)nd * RTX_WIDE + s];
                    uint32_t kind = c.ref & RTX_WREF_KIND;
                    if (kind == RTX_WREF_EMPTY) continue;
                    bool any = false;
                    double tn_min = 1e300;
                    for (int k = 0; k < 16; ++k)
                        if (seg_box(seg[k], c.lo, c.hi, centre)) { any = true; tn_min = std::min(tn_min, g_tn); }   // the beam is the hull of ALL 16 (dead ones too)
                    if (!any) continue;
                    if (kind == RTX_WREF_NODE) {
                        int below = 0;
                        for (int q = 0; q < s; ++q) below += (wide[(size_t)nd * RTX_WIDE + q].ref & RTX_WREF_KIND) == RTX_WREF_NODE;
                        kids.push_back(c.first_node + (uint32_t)below);
                        kid_tn.push_back(tn_min);
                    } else {
                        s_queued += 1;
                        for (int k = 0; k < 16; ++k) {
                            if (!seg[k].alive) continue;
                            s_box += 1;
                            if (!seg_box(seg[k], c.lo, c.hi, centre)) continue;
                            s_exact += 1;
                            if (seg_tri(seg[k], &tv[(size_t)(c.ref & RTX_WREF_INDEX) * 9])) { seg[k].alive = false; --live; }
                        }
                    }
                }
                static const int order_mode = std::getenv("BVH_ORDER") ? std::atoi(std::getenv("BVH_ORDER")) : 0;
                if (order_mode == 1) {   // nearest child first (all children sorted by entry distance)
                    std::vector<size_t> ix(kids.size());
                    for (size_t q = 0; q < ix.size(); ++q) ix[q] = q;
                    std::sort(ix.begin(), ix.end(), [&](size_t a, size_t b) { return kid_tn[a] < kid_tn[b]; });
                    std::vector<uint32_t> k2;
                    for (size_t q : ix) k2.push_back(kids[q]);
                    kids = k2;
                } else if (order_mode == 2 && kids.size() > 1) {   // only the nearest child moved to the front
                    size_t m = 0;
                    for (size_t q = 1; q < kids.size(); ++q) if (kid_tn[q] < kid_tn[m]) m = q;
                    std::swap(kids[0], kids[m]);
                }
                for (size_t q = kids.size(); q-- > 0;) { stack.push_back(kids[q]); dstack.push_back(dep + 1); }
            }
            s_nodes += visits;
            s_steps += (visits + 1) / 2;
            s_occl += 16 - live;
        }
        std::printf("light %lld: node visits per group %.2f (two-node steps %.2f), prims queued per group %.2f, own-box tests per feeler %.2f, "
                    "exact tests per feeler %.2f, occluded %.3f\n", (long long)li, s_nodes / groups, s_steps / groups, s_queued / groups,
                    s_box / groups / 16, s_exact / groups / 16, s_occl / groups / 16);
        std::printf("  visits by depth:");
        for (int d = 0; d < 10; ++d) std::printf(" %.2f", by_depth[d] / groups);
        std::printf("\n  visits by fill:");
        for (int d = 1; d <= 16; ++d) std::printf(" %.2f", by_fill[d] / groups);
        std::printf("\n");
    }
    return 0;
}
