// Host harness for vqvae_amd/csrc/vq_track.h (test infrastructure; built by tests/test_vq_track_host.py with g++).
// Emulates the two accumulator lanes (halves) of every row over a synthetic screen matrix acc[row][code] exactly as the
// kernel walks it (tile by tile, register r <-> code 32 T + (r & 3) + 8 (r >> 2) + 4 h) and checks the tracker's verdict
// against a brute-force scan:
//   closed rows: exactly one code is at or above the threshold, and kbest is that code;
//   open rows:   the union of the exact tasks' codes contains every code at or above the threshold, all codes < K;
//   hard rows:   only when a half really has more than two streams / three or more cell keys at or above the threshold
//                i.e. the rescan is not taken spuriously.
// usage: trk_harness N K seed mode   (mode 0: gaussian scores, 1: near ties, 2: exact duplicates, 3: negative / tiny values)
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <random>
#include <set>
#include <vector>
#include "../../vqvae_amd/csrc/vq_track.h"

using namespace vqvae::trk;

int main(int argc, char **argv) {
    if (argc < 5) return 2;
    const int N = atoi(argv[1]), K = atoi(argv[2]), seed = atoi(argv[3]), mode = atoi(argv[4]);
    const int K32 = (K + 31) / 32 * 32, ntile = K32 / 32;
    std::mt19937 rng(seed);
    std::normal_distribution<float> nd(0.0f, 1.0f);
    std::uniform_real_distribution<float> ud(0.0f, 1.0f);
    std::uniform_int_distribution<int> uk(0, K - 1);
    const float ninf = -INFINITY, pinf = INFINITY;
    long closed = 0, open = 0, hard = 0, tasks = 0, viol = 0, single_not_closed = 0;
    std::vector<float> acc(K32);
    for (int n = 0; n < N; ++n) {
        const float scale = mode == 3 ? 1e-3f : 100.0f;
        const float shift = mode == 3 ? -5.0f * scale : 0.0f;
        for (int k = 0; k < K32; ++k) acc[k] = k < K ? nd(rng) * scale + shift : -3.0e38f;
        float mx = -INFINITY;
        for (int k = 0; k < K; ++k) mx = std::fmax(mx, acc[k]);
        float mag = 0.0f;
        for (int k = 0; k < K; ++k) mag = std::fmax(mag, std::fabs(acc[k]));
        const float delta = 0.02f * scale * (0.2f + ud(rng));
        if (mode == 1 || mode == 2) {
            // plant up to 6 more codes within delta of the maximum (mode 2: bit-identical to it)
            const int extra = (int)(ud(rng) * 7.0f);
            for (int e = 0; e < extra; ++e) {
                const int k = uk(rng);
                acc[k] = mode == 2 ? mx : mx - delta * ud(rng) * 1.2f;
            }
            mx = -INFINITY;
            for (int k = 0; k < K; ++k) mx = std::fmax(mx, acc[k]);
        }
        const float thr = mx - delta;
        const float thrB = thr - 8.0e-6f * (mag * 1.01f + 1e-30f);
        std::set<int> G;
        for (int k = 0; k < K; ++k) if (acc[k] >= thr) G.insert(k);
        // the two lanes of the row
        Lane L[2];
        for (int h = 0; h < 2; ++h) {
            init(L[h], ninf);
            for (int T = 0; T < ntile; ++T) {
                float a[16];
                for (int r = 0; r < 16; ++r) a[r] = acc[32 * T + (r & 3) + 8 * (r >> 2) + 4 * h];
                tile(L[h], a, 2u * T, 2u * T + 1u, kKeyMask, ninf, pinf);
            }
        }
        const float v1 = std::fmax(lane_max(L[0], ninf), lane_max(L[1], ninf));
        if (v1 != mx) { ++viol; fprintf(stderr, "row %d: v1 %g != max %g\n", n, v1, mx); }
        Half H[2] = {half_of(L[0], thr, thrB, 0), half_of(L[1], thr, thrB, 1)};
        const Verdict V0 = verdict_of(H[0], pack(H[1]), K), V1 = verdict_of(H[1], pack(H[0]), K);
        if (V0.closed != V1.closed || V0.hard != V1.hard || (V0.closed && V0.kbest != V1.kbest)) {
            ++viol; fprintf(stderr, "row %d: halves disagree\n", n);
        }
        // round 5: the word / speaker form must take the same decisions, and its tasks must cover the same codes
        {
            const unsigned ge0 = ~lt_of(L[0], thr, thrB, ninf) & kGeBits, ge1 = ~lt_of(L[1], thr, thrB, ninf) & (kGeBits | 1u << 14);
            const Spoken S = spoken_of(word_of(L[0], ge0), word_of(L[1], ge1));
            const bool c2 = spoken_closed(S, K), h2 = !c2 && spoken_hard(S, K);
            if (c2 != V0.closed || h2 != V0.hard || (c2 && S.kbest != V0.kbest)) {
                ++viol; fprintf(stderr, "row %d: speaker form disagrees (closed %d/%d hard %d/%d kbest %d/%d)\n", n, (int)c2, (int)V0.closed, (int)h2, (int)V0.hard, S.kbest, V0.kbest);
            }
            if (!c2 && !h2) {
                const Cands A[2] = {cands_of(L[0], H[0], 0, K), cands_of(L[1], H[1], 1, K)};
                const Cands B[2] = {cands2_of(L[0], ge0, 0, K), cands2_of(L[1], ge1, 1, K)};
                for (int h = 0; h < 2; ++h) {                   // (the two forms number a half's streams in opposite orders)
                    std::set<int> sa, sb;
                    for (int j = 0; j < A[h].ntask; ++j) { sa.insert(A[h].ta[j]); sa.insert(A[h].tb[j]); }
                    for (int j = 0; j < B[h].ntask; ++j) { sb.insert(B[h].ta[j]); sb.insert(B[h].tb[j]); }
                    if (A[h].ntask != B[h].ntask || sa != sb) { ++viol; fprintf(stderr, "row %d half %d: tasks differ\n", n, h); }
                }
            }
        }
        if (V0.closed) {
            ++closed;
            if (G.size() != 1 || *G.begin() != V0.kbest) { ++viol; fprintf(stderr, "row %d: closed with |G| = %zu, kbest %d\n", n, G.size(), V0.kbest); }
            continue;
        }
        if (G.size() == 1) ++single_not_closed;
        bool is_hard = V0.hard;
        std::set<int> cand;
        if (!is_hard) {
            Cands C[2] = {cands_of(L[0], H[0], 0, K), cands_of(L[1], H[1], 1, K)};
            for (int h = 0; h < 2; ++h)
                    for (int j = 0; j < C[h].ntask; ++j) { cand.insert(C[h].ta[j]); cand.insert(C[h].tb[j]); ++tasks; }
        }
        if (is_hard) {
            ++hard;
            // spurious? brute-force per-half counts
            bool need = false;
            for (int h = 0; h < 2; ++h) {
                int sA = 0, cB = 0;
                for (int a = 0; a < 8; ++a) {
                    bool any = false;
                    for (int T = 0; T < ntile; ++T) for (int s = 0; s < 2; ++s) any |= acc[code_of(a, 2 * T + s, h)] >= thr;
                    sA += any;
                }
                for (int c = 0; c < 2 * ntile; ++c) {
                    float x = -INFINITY;
                    for (int a = 0; a < 8; ++a) x = std::fmax(x, acc[code_of(a, c, h)]);
                    cB += u2f((f2u(x) & kKeyMask) | (unsigned)c) >= thrB;
                }
                need |= sA > 2 || cB > 2;
            }
            if (!need) { ++viol; fprintf(stderr, "row %d: spurious hard verdict\n", n); }
            continue;
        }
        ++open;
        for (int k : G) if (!cand.count(k)) { ++viol; fprintf(stderr, "row %d: code %d >= thr not among the tasks\n", n, k); }
        for (int k : cand) if (k < 0 || k >= K) { ++viol; fprintf(stderr, "row %d: task code %d out of range\n", n, k); }
        if (cand.empty()) { ++viol; fprintf(stderr, "row %d: open without tasks\n", n); }
    }
    printf("{\"rows\": %d, \"closed\": %ld, \"open\": %ld, \"hard\": %ld, \"tasks\": %ld, \"single_not_closed\": %ld, \"violations\": %ld}\n",
           N, closed, open, hard, tasks, single_not_closed, viol);
    return viol ? 1 : 0;
}
