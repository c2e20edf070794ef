/*
 * sim_sync.c -- TOOL (not product, not a test): a CPU model of the speculative entropy kernel's synchronisation
 * (amv_huffman_sync2_kernel, csrc/amv_decode_sync.hip) on the synthetic stream, to see how many re-walk rounds frames need
 * and what the states at a moving "front" look like.  Uses the oracle's tables (test infrastructure) by including its
 * source.
 *
 *     gcc -O2 -fopenmp -o tools/scratch/sim_sync tools/sim_sync.c -lm && tools/scratch/sim_sync [frames] [lanes] [policy]
 */
#include "../oracle/amv_oracle.c"
#include <stdio.h>

typedef struct { uint32_t t, k, j; } st3;      /* t: bits consumed; k: next index (0 = DC next); j: block in MCU 0..5 */

static hufbounds g_hb[4];
static uint64_t g_walks;
static int g_cap = 48, g_reserve = 0;

/* one symbol of the skip walk at bit position *t of bits[] (zero beyond nbits); returns 0 or 1 = "invalid code: slipped a bit" */
static inline unsigned getbits(const uint8_t *bits, uint32_t nbits, uint32_t t, int n)
{
    unsigned v = 0;
    for (int i = 0; i < n; i++) { v <<= 1; if (t + i < nbits) v |= bits[t + i]; }
    return v;
}

static int skip_symbol(const uint8_t *bits, uint32_t nbits, st3 *s, int *anomaly)
{
    const int chroma = s->j >= 4, tab = (s->k ? 2 : 0) + chroma;
    const hufbounds *hb = &g_hb[tab];
    int code = 0, len = 0, found = 0, sym = 0;
    uint32_t t = s->t;
    while (len < 16) {
        code = (code << 1) | (int)getbits(bits, nbits, t + len, 1);
        len++;
        if (hb->cnt[len - 1] && code >= hb->minc[len - 1] && code <= hb->maxc[len - 1]) { found = 1; break; }
    }
    if (!found) { s->t += 1; *anomaly = 1; return 1; }     /* slip one bit, leave the index */
    sym = k_vals[tab][(uint16_t)(code - hb->minc[len - 1] + hb->pos[len - 1])];
    const int run = sym >> 4, size = sym & 15;
    s->t += len + size;
    uint32_t kn;
    int end = 0;
    if (s->k == 0) kn = 1;
    else if (run == 0 && size == 0) { end = 1; kn = 64; }
    else { kn = s->k + run + 1; if (kn > 64) *anomaly = 2; if (kn >= 64) end = 1; }
    if (end) { s->k = 0; s->j = s->j == 5 ? 0 : s->j + 1; } else s->k = kn;
    return 0;
}

static st3 walk(const uint8_t *bits, uint32_t nbits, st3 s, uint32_t limit, uint32_t *nanom)
{
    int a;
    while (s.t < limit) { a = 0; skip_symbol(bits, nbits, &s, &a); if (a && nanom) (*nanom)++; }
    return s;
}

/* policy 2: Jacobi rounds as the kernel has them + a memo per lane (entries it has walked from, with the arrival) + a
 * finality cascade from lane 0 at the start of every round: a lane whose left neighbour is final takes that neighbour's
 * arrival as its (final) entry; when it has walked from it before, its arrival is final too, in the same round, and so on
 * to the right.  Final lanes never change again.  Returns the number of rounds in which some lane walked (the first walk
 * not counted) and in *walks the lane-walks made. */
#define MEMO 8
typedef struct { st3 e[MEMO], a[MEMO]; int n; } memo_t;
static int same(st3 a, st3 b) { return a.t == b.t && a.k == b.k && a.j == b.j; }
static int memo_find(const memo_t *m, st3 e, int depth)
{
    int lo = m->n - depth;
    if (lo < 0) lo = 0;
    for (int q = m->n - 1; q >= lo; q--) if (same(m->e[q], e)) return q;
    return -1;
}
static void memo_add(memo_t *m, st3 e, st3 a)
{
    if (m->n == MEMO) { memmove(m->e, m->e + 1, sizeof(st3) * (MEMO - 1)); memmove(m->a, m->a + 1, sizeof(st3) * (MEMO - 1)); m->n--; }
    m->e[m->n] = e; m->a[m->n] = a; m->n++;
}
static uint32_t memo_rounds(const uint8_t *bits, uint32_t nb, uint32_t L, uint32_t S, const st3 *truth, int depth, uint32_t *walks, int jacobi)
{
    memo_t memo[64];
    st3 entry[64], arrive[64];
    int fin[64] = {0};          /* 0 not final, 1 entry final (arrival not yet known), 2 entry and arrival final */
    uint32_t rounds = 0;
    for (uint32_t i = 0; i < L; i++) {
        memo[i].n = 0;
        entry[i] = i ? (st3){i * S, 0, 0} : (st3){0, 0, 0};
        arrive[i] = i + 1 < L ? walk(bits, nb + 64, entry[i], (i + 1) * S, NULL) : entry[i];
        memo_add(&memo[i], entry[i], arrive[i]);
    }
    fin[0] = 2;
    for (uint32_t r = 0; r < 4 * L; r++) {
        int need[64] = {0}, any = 0, allfin = 1;
        st3 next_entry[64];
        /* finality cascade */
        for (uint32_t i = 1; i < L; i++) {
            if (fin[i] == 2) continue;
            if (fin[i - 1] != 2) break;
            entry[i] = arrive[i - 1];
            const int q = memo_find(&memo[i], entry[i], depth);
            if (q >= 0) { arrive[i] = memo[i].a[q]; fin[i] = 2; }
            else { fin[i] = 1; need[i] = 1; break; }
        }
        /* the others: Jacobi on the arrivals of the round before */
        for (uint32_t i = 1; i < L; i++) next_entry[i] = arrive[i - 1];
        for (uint32_t i = 1; i < L; i++) {
            if (fin[i]) continue;
            if (!jacobi) continue;
            if (same(next_entry[i], entry[i])) continue;
            entry[i] = next_entry[i];
            const int q = memo_find(&memo[i], entry[i], depth);
            if (q >= 0) arrive[i] = memo[i].a[q];
            else need[i] = 1;
        }
        for (uint32_t i = 0; i < L; i++) { if (need[i]) any = 1; if (fin[i] != 2) allfin = 0; }
        if (!any) { if (allfin) break; else continue; }       /* (a cascade that found everything: next trip continues it) */
        rounds++;
        if (getenv("SIM_TRACE2")) {
            printf("   round %u: fin", rounds);
            for (uint32_t i = 0; i < L; i++) printf("%d", fin[i]);
            printf(" walk");
            for (uint32_t i = 0; i < L; i++) printf("%d", need[i]);
            printf(" true-entry");
            for (uint32_t i = 0; i < L; i++) printf("%d", same(entry[i], truth[i]));
            printf("\n");
            if (rounds >= 3)
                for (uint32_t i = 1; i < L; i++)
                    if (!same(entry[i], truth[i])) printf("        lane %u has (%u,%u,%u) truth (%u,%u,%u)\n", i, entry[i].t - i * S, entry[i].k, entry[i].j, truth[i].t - i * S, truth[i].k, truth[i].j);
        }
        for (uint32_t i = 1; i < L; i++)
            if (need[i]) {
                arrive[i] = i + 1 < L ? walk(bits, nb + 64, entry[i], (i + 1) * S, NULL) : entry[i];
                memo_add(&memo[i], entry[i], arrive[i]);
                if (fin[i] == 1) fin[i] = 2;
                (*walks)++;
            }
    }
    for (uint32_t i = 0; i < L; i++)
        if (!same(entry[i], truth[i])) { printf("memo policy: lane %u is not the truth\n", i); break; }
    return rounds;
}

/* policy 5: a pool of (share, entry) tasks per frame, L walks per round.  memo[x]: entries share x has been walked from, with
 * the arrivals; every arrival of a walk of share x - 1 is a candidate entry for share x; seeds: fresh guesses (k = 0) with
 * the phases the first walk did not try.  The finality cascade runs at the start of every round. */
#define BIGMEMO 48
typedef struct { st3 e[BIGMEMO], a[BIGMEMO]; int n; } bigmemo_t;
static int big_find(const bigmemo_t *m, st3 e) { for (int q = 0; q < m->n; q++) if (same(m->e[q], e)) return q; return -1; }
static uint32_t pool_rounds(const uint8_t *bits, uint32_t nb, uint32_t L, uint32_t S, const st3 *truth, uint32_t *walks, int seeds_on, int *maxmemo)
{
    static bigmemo_t memo[64], cand[64];
    int seeded[64] = {0};
    st3 fin_entry[64];
    uint32_t nfinal = 1, rounds = 0;         /* lanes [0, nfinal) have final entries */
    for (uint32_t i = 0; i < L; i++) { memo[i].n = 0; cand[i].n = 0; }
    fin_entry[0] = (st3){0, 0, 0};
    for (uint32_t i = 1; i < L; i++) { cand[i].e[0] = (st3){i * S, 0, 0}; cand[i].n = 1; }
    cand[0].e[0] = fin_entry[0]; cand[0].n = 1;
    for (uint32_t r = 0; r < 8 * L; r++) {
        /* cascade: lane nfinal's entry = arrival of lane nfinal - 1 from ITS final entry, when that is known */
        while (nfinal < L) {
            const int q = big_find(&memo[nfinal - 1], fin_entry[nfinal - 1]);
            if (q < 0) break;
            fin_entry[nfinal] = memo[nfinal - 1].a[q];
            nfinal++;
        }
        if (nfinal == L) break;      /* (the last share is never walked: every entry is final) */
        /* tasks */
        struct { uint32_t x; st3 e; } task[64];
        uint32_t nt = 0;
        task[nt].x = nfinal - 1; task[nt].e = fin_entry[nfinal - 1]; nt++;        /* the front: its final entry is not in its memo */
        for (uint32_t x = nfinal; x + 1 < L && nt < L; x++)
            for (int c = cand[x].n - 1; c >= 0 && nt < L; c--)
                if (big_find(&memo[x], cand[x].e[c]) < 0) {
                    int dup = 0;
                    for (uint32_t q = 0; q < nt; q++) if (task[q].x == x && same(task[q].e, cand[x].e[c])) dup = 1;
                    if (!dup) { task[nt].x = x; task[nt].e = cand[x].e[c]; nt++; }
                }
        if (seeds_on && r >= 1)
            for (uint32_t x = nfinal; x + 1 < L && nt < L; x++)
                while (seeded[x] < 5 && nt < L && memo[x].n + seeded[x] < g_cap - g_reserve) { seeded[x]++; task[nt].x = x; task[nt].e = (st3){x * S, 0, (uint32_t)seeded[x]}; nt++; }
        rounds++;
        for (uint32_t q = 0; q < nt; q++) {
            const uint32_t x = task[q].x;
            const st3 a = walk(bits, nb + 64, task[q].e, (x + 1) * S, NULL);
            (*walks)++;
            if (memo[x].n < g_cap) { memo[x].e[memo[x].n] = task[q].e; memo[x].a[memo[x].n] = a; memo[x].n++; }
            else if (x == nfinal - 1 && q == 0) { memo[x].e[g_cap - 1] = task[q].e; memo[x].a[g_cap - 1] = a; }   /* the front's walk always lands */
            if (memo[x].n > *maxmemo) *maxmemo = memo[x].n;
            if (x + 1 < L && big_find(&cand[x + 1], a) < 0 && cand[x + 1].n < BIGMEMO) { cand[x + 1].e[cand[x + 1].n++] = a; }
        }
    }
    for (uint32_t i = 0; i < L; i++)
        if (!same(fin_entry[i], truth[i])) { printf("pool policy: lane %u is not the truth\n", i); break; }
    return rounds - 1;          /* as the other policies count: rounds behind the first walk */
}

/* policy 7: the kernel's own round (amv_huffman_sync2_kernel, round 5), lane for lane: every share keeps a memo of g_cap
 * walks; per round a share lists its final start or else up to g_pend candidates (arrivals in the left share's memo it has
 * not walked from, newest first) and seeds; a share's first task is its own lane's, the others go to lanes without a task,
 * in lane order. */
static int g_pend = 4, g_seedres = 1, g_variant = 0, g_front_only_from = 0, g_seeds = 5;
static uint32_t kernel_rounds(const uint8_t *bits, uint32_t nb, uint32_t L, uint32_t S, const st3 *truth, uint32_t *walks)
{
    static bigmemo_t memo[64];
    int fin[64] = {0}, seeded[64] = {0};
    st3 fin_code[64], fin_arr[64];
    uint32_t rounds = 0;
    for (uint32_t i = 0; i < L; i++) {
        memo[i].n = 0;
        if (i + 1 < L) {
            st3 e = {i * S, 0, 0};
            memo[i].e[0] = e; memo[i].a[0] = walk(bits, nb + 64, e, (i + 1) * S, NULL); memo[i].n = 1;
        }
    }
    fin[0] = 1; fin_code[0] = (st3){0, 0, 0};
    for (uint32_t r = 0; r < L + 1; r++) {
        for (uint32_t hop = 0; hop < L; hop++) {
            int moved = 0;
            int ld[64]; st3 la[64];
            for (uint32_t i = 1; i < L; i++) { ld[i] = fin[i - 1] == 2; la[i] = fin_arr[i - 1]; }
            for (uint32_t i = 0; i < L; i++) {
                if (fin[i] == 0 && i && ld[i]) { fin[i] = 1; fin_code[i] = la[i]; moved = 1; }
                if (fin[i] == 1) {
                    if (i == L - 1) { fin[i] = 2; moved = 1; }
                    else { const int q = big_find(&memo[i], fin_code[i]); if (q >= 0) { fin[i] = 2; fin_arr[i] = memo[i].a[q]; moved = 1; } }
                }
            }
            if (!moved) break;
        }
        int open = 0;
        for (uint32_t i = 0; i < L; i++) if (fin[i] != 2) open = 1;
        if (!open) break;
        rounds++;
        if (getenv("SIM_TRACE3")) {
            printf("   r%u fin ", rounds);
            for (uint32_t i = 0; i < L; i++) printf("%d", fin[i]);
            printf(" memo ");
            for (uint32_t i = 0; i < L; i++) printf("%d", memo[i].n);
            printf(" true-in-memo ");
            for (uint32_t i = 0; i + 1 < L; i++) printf("%d", big_find(&memo[i], truth[i]) >= 0);
            printf("\n");
        }
        st3 pend[64][8];
        uint32_t p[64] = {0};
        for (uint32_t i = 0; i + 1 < L; i++) {
            if (fin[i] == 1) { pend[i][0] = fin_code[i]; p[i] = 1; }
            else if (fin[i] == 0 && !(g_front_only_from && r + 1 >= (uint32_t)g_front_only_from)) {
                const int room = g_cap - memo[i].n;
                const uint32_t most = (uint32_t)(room < g_pend ? (room < 0 ? 0 : room) : g_pend);
                for (int q = memo[i - 1].n - 1; q >= 0; q--) {
                    const st3 c = memo[i - 1].a[q];
                    int known = big_find(&memo[i], c) >= 0;
                    for (uint32_t m = 0; m < p[i]; m++) known |= same(pend[i][m], c);
                    if (!known && p[i] < most) pend[i][p[i]++] = c;
                }
                if (g_variant == 0)
                    while (seeded[i] < 5 && p[i] + g_seedres < most) { seeded[i]++; pend[i][p[i]++] = (st3){i * S, 0, (uint32_t)seeded[i]}; }
                else if (g_variant == 4)      /* more seeds: the six block positions at bit offsets 0, 3, 6, ... of the share's start */
                    while (seeded[i] < g_seeds && p[i] + g_seedres < most) {
                        seeded[i]++;
                        pend[i][p[i]++] = (st3){i * S + 3u * (uint32_t)(seeded[i] / 6), 0, (uint32_t)(seeded[i] % 6)};
                    }
                else if (memo[i - 1].n) {
                    /* phase variants of an arrival at the end of the share before: variant 1 the newest, 2 the oldest (the guess's) */
                    const st3 c0 = memo[i - 1].a[g_variant == 1 ? memo[i - 1].n - 1 : 0];
                    while (seeded[i] < 5 && p[i] + g_seedres < most) {
                        seeded[i]++;
                        const st3 v = {c0.t, c0.k, (c0.j + (uint32_t)seeded[i]) % 6};
                        int known = big_find(&memo[i], v) >= 0;
                        for (uint32_t m = 0; m < p[i]; m++) known |= same(pend[i][m], v);
                        if (!known) pend[i][p[i]++] = v;
                    }
                }
            }
        }
        uint32_t idle_lanes[64], ni = 0;
        for (uint32_t i = 0; i < L; i++) if (p[i] == 0) idle_lanes[ni++] = i;
        uint32_t used = 0;
        for (uint32_t i = 0; i + 1 < L; i++) {
            uint32_t done = 0;
            for (uint32_t m = 0; m < p[i]; m++) {
                if (m > 0) { if (used >= ni) break; used++; }
                const st3 a = walk(bits, nb + 64, pend[i][m], (i + 1) * S, NULL);
                (*walks)++;
                if (m == 0 && fin[i] == 1) { fin[i] = 2; fin_arr[i] = a; }
                if (memo[i].n + (int)done < g_cap) { memo[i].e[memo[i].n + done] = pend[i][m]; memo[i].a[memo[i].n + done] = a; }
                done++;
            }
            memo[i].n = memo[i].n + (int)done < g_cap ? memo[i].n + (int)done : g_cap;
        }
    }
    for (uint32_t i = 0; i < L; i++)
        if (fin[i] != 2 || !same(fin_code[i], truth[i])) { printf("kernel policy: lane %u is not the truth (fin %d)\n", i, fin[i]); break; }
    return rounds;
}

int main(int argc, char **argv)
{
    const uint32_t n = argc > 1 ? (uint32_t)atoi(argv[1]) : 2000, L = argc > 2 ? (uint32_t)atoi(argv[2]) : 16;
    const int policy = argc > 3 ? atoi(argv[3]) : 0;
    const uint32_t w = argc > 4 ? (uint32_t)atoi(argv[4]) : 160, h = argc > 5 ? (uint32_t)atoi(argv[5]) : 120;
    if (getenv("SIM_CAP")) g_cap = atoi(getenv("SIM_CAP"));
    if (getenv("SIM_RESERVE")) g_reserve = atoi(getenv("SIM_RESERVE"));
    if (getenv("SIM_PEND")) g_pend = atoi(getenv("SIM_PEND"));
    if (getenv("SIM_SEEDRES")) g_seedres = atoi(getenv("SIM_SEEDRES"));
    if (getenv("SIM_VARIANT")) g_variant = atoi(getenv("SIM_VARIANT"));
    if (getenv("SIM_FRONT_ONLY")) g_front_only_from = atoi(getenv("SIM_FRONT_ONLY"));
    if (getenv("SIM_SEEDS")) g_seeds = atoi(getenv("SIM_SEEDS"));
    uint64_t cap = (uint64_t)n * w * h;
    uint8_t *blob = malloc(cap);
    uint64_t *offs = malloc(n * 8);
    uint32_t *lens = malloc(n * 4);
    for (int i = 0; i < 4; i++) build_bounds(&g_hb[i], k_bits[i]);
    amvo_synth_encode_batch(0xA11CE, 0, n, w, h, 0, blob, cap, offs, lens, 8);
    uint64_t hist_rounds[80] = {0}, tot_rounds = 0, pure = 0, fronts = 0, same_tk = 0, hist_w[64] = {0};
    uint64_t waves_rounds = 0, nwaves = 0, wave_hist[80] = {0};
    uint32_t wave_max = 0, in_wave = 0, frames_per_wave = 64 / L;
    for (uint32_t f = 0; f < n; f++) {
        const uint8_t *c = blob + offs[f];
        uint32_t len = lens[f], nb = 0;
        uint8_t *bits = malloc((size_t)len * 8 + 64);
        for (uint32_t i = 2; i < len; i++) {           /* unstuff: drop the byte behind each FF */
            uint8_t b = c[i];
            for (int q = 7; q >= 0; q--) bits[nb++] = (b >> q) & 1;
            if (b == 0xff) i++;
        }
        const uint32_t total_bytes = nb / 8, valid_bits = total_bytes * 8;
        uint32_t S = ((valid_bits + L - 1) / L + 31u) & ~31u;
        if (S < 64) S = 64;
        st3 entry[64], arrive[64], old_entry[64];
        uint32_t rounds = 0;
        /* true states at share boundaries, for analysis */
        st3 truth[65];
        {
            st3 s = {0, 0, 0};
            for (uint32_t i = 0; i < L; i++) { truth[i] = s; s = walk(bits, nb + 64, s, (i + 1) * S, NULL); }
        }
        if (getenv("SIM_FRAME") && (uint32_t)atoi(getenv("SIM_FRAME")) == f) {
            st3 s2 = {0, 0, 0};
            printf("frame %u: %u bits, S %u\n", f, nb, S);
            for (uint32_t i = 0; i < L; i++) {
                uint32_t nsym = 0, nblk = 0, mcu0 = 0;
                while (s2.t < (i + 1) * S && s2.t < nb) { int a = 0; const uint32_t j0 = s2.j; skip_symbol(bits, nb + 64, &s2, &a); nsym++; if (s2.j != j0) nblk++; }
                (void)mcu0;
                printf("  share %u: %u symbols, %u blocks (%.1f bits per symbol, %.0f bits per block)\n", i, nsym, nblk, (double)S / nsym, nblk ? (double)S / nblk : 0.0);
            }
        }
        for (uint32_t i = 0; i < L; i++) {
            entry[i] = (st3){i * S, 0, 0};
            arrive[i] = i + 1 < L ? walk(bits, nb + 64, entry[i], (i + 1) * S, NULL) : entry[i];
        }
        /* merge distance of each lane's first walk with the truth: count of shares until arrival state == truth */
        for (uint32_t i = 1; i + 1 < L; i++) {
            st3 s = entry[i];
            uint32_t sh = 0;
            for (uint32_t q = i; q < L; q++) {
                s = walk(bits, nb + 64, s, (q + 1) * S, NULL);
                sh++;
                if (q + 1 < L && s.t == truth[q + 1].t && s.k == truth[q + 1].k && s.j == truth[q + 1].j) break;
            }
            hist_w[sh < 63 ? sh : 63]++;
        }
        uint32_t nwalks = 0;
        static int maxmemo;
        if (policy == 7) {
            rounds = kernel_rounds(bits, nb, L, S, truth, &nwalks);
            if (getenv("SIM_TRACE3")) printf("frame %u: %u rounds\n", f, rounds);
            g_walks += nwalks;
        } else if (policy == 5 || policy == 6) {
            rounds = pool_rounds(bits, nb, L, S, truth, &nwalks, policy == 6, &maxmemo);
            g_walks += nwalks;
            if (f == n - 1) printf("largest memo: %d entries\n", maxmemo);
        } else if (policy >= 2) {
            rounds = memo_rounds(bits, nb, L, S, truth, policy / 10 ? policy / 10 : MEMO, &nwalks, (policy % 10) == 2);
            if (getenv("SIM_TRACE2")) printf("frame %u: %u rounds\n", f, rounds);
            g_walks += nwalks;
        }
        for (uint32_t r = 0; r < L && policy < 2; r++) {
            int any = 0;
            st3 na[64];
            int changed[64] = {0};
            for (uint32_t i = 1; i < L; i++) {
                const st3 l = arrive[i - 1];
                if (l.t != entry[i].t || l.k != entry[i].k || l.j != entry[i].j) { changed[i] = 1; any = 1; }
            }
            if (!any) break;
            rounds++;
            int shift = -1;      /* phase shift seen at the leftmost pure-phase front of this round */
            uint32_t front = 0;
            for (uint32_t i = 1; i < L; i++) {
                if (!changed[i]) continue;
                const st3 l = arrive[i - 1];
                if (r >= 1) {
                    fronts++;
                    if (l.t == entry[i].t && l.k == entry[i].k) { same_tk++; }
                    if (l.t == entry[i].t && l.k == entry[i].k && l.j != entry[i].j) {
                        pure++;
                        if (shift < 0) { shift = (int)((l.j + 6 - entry[i].j) % 6); front = i; }
                    }
                }
            }
            if (getenv("SIM_TRACE") && r >= 3)
                for (uint32_t i = 1; i < L; i++)
                    if (changed[i]) printf("  f%u r%u lane %u: entry (%u,%u,%u) -> (%u,%u,%u)  truth (%u,%u,%u)\n", f, r, i, entry[i].t - i * S, entry[i].k, entry[i].j,
                                           arrive[i - 1].t - i * S, arrive[i - 1].k, arrive[i - 1].j, truth[i].t - i * S, truth[i].k, truth[i].j);
            for (uint32_t i = 0; i < L; i++) { na[i] = arrive[i]; old_entry[i] = entry[i]; }
            for (uint32_t i = 1; i < L; i++) {
                if (changed[i]) {
                    entry[i] = arrive[i - 1];
                    na[i] = i + 1 < L ? walk(bits, nb + 64, entry[i], (i + 1) * S, NULL) : entry[i];
                    g_walks++;
                } else if (policy == 1 && shift > 0 && i > front) {
                    /* lanes right of a pure phase front shift their entry's phase by the same amount and re-walk */
                    entry[i].j = (entry[i].j + (uint32_t)shift) % 6;
                    na[i] = i + 1 < L ? walk(bits, nb + 64, entry[i], (i + 1) * S, NULL) : entry[i];
                }
            }
            for (uint32_t i = 0; i < L; i++) arrive[i] = na[i];
        }
        /* check the fixed point is the truth */
        for (uint32_t i = 0; i < L && policy < 2; i++)
            if (entry[i].t != truth[i].t || entry[i].k != truth[i].k || entry[i].j != truth[i].j) { printf("frame %u lane %u: fixed point is not the truth\n", f, i); break; }
        hist_rounds[rounds]++;
        tot_rounds += rounds;
        if (rounds > wave_max) wave_max = rounds;
        if (++in_wave == frames_per_wave) { waves_rounds += wave_max; wave_hist[wave_max]++; nwaves++; wave_max = 0; in_wave = 0; }
        free(bits);
    }
    printf("frames %u lanes %u policy %d: mean rounds %.3f; per wave of %u frames: mean of max %.3f\n", n, L, policy, (double)tot_rounds / n, frames_per_wave,
           (double)waves_rounds / (nwaves ? nwaves : 1));
    printf("lane-walks behind the first: %.2f per frame\n", (double)g_walks / n);
    printf("rounds histogram (frames):");
    for (int i = 0; i < 40; i++) if (hist_rounds[i]) printf(" %d:%llu", i, (unsigned long long)hist_rounds[i]);
    printf("\nrounds histogram (waves): ");
    for (int i = 0; i < 40; i++) if (wave_hist[i]) printf(" %d:%llu", i, (unsigned long long)wave_hist[i]);
    printf("\nfronts after round 1: %llu, same (t,k): %llu, pure phase shifts: %llu\n", (unsigned long long)fronts, (unsigned long long)same_tk, (unsigned long long)pure);
    printf("first-walk merge distance in shares:");
    for (int i = 0; i < 64; i++) if (hist_w[i]) printf(" %d:%llu", i, (unsigned long long)hist_w[i]);
    printf("\n");
    return 0;
}
