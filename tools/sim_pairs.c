/* TOOL (not product, not a test), like sim_sync.c: how many look-up steps would a walk need if one look-up could take TWO symbols whose bits (code + magnitude)
 * fit the table's index together?  Uses the oracle's tables like tools/sim_sync.c. */
#include "../oracle/amv_oracle.c"
#include <stdio.h>
typedef struct { uint32_t t, k, j; } st3;
static hufbounds g_hb[4];
static inline unsigned getbit(const uint8_t *bits, uint32_t nbits, uint32_t t) { return t < nbits ? bits[t] : 0; }
/* returns total bits of the symbol; *end = block ended; *dc = was a DC symbol; *val = carries a value */
static int sym(const uint8_t *bits, uint32_t nb, st3 *s, int *end, int *dc, int *val)
{
    const int chroma = s->j >= 4, tab = (s->k ? 2 : 0) + chroma;
    const hufbounds *hb = &g_hb[tab];
    int code = 0, len = 0, found = 0;
    while (len < 16) {
        code = (code << 1) | (int)getbit(bits, nb, s->t + len);
        len++;
        if (hb->cnt[len - 1] && code >= hb->minc[len - 1] && code <= hb->maxc[len - 1]) { found = 1; break; }
    }
    if (!found) return -1;
    const int sy = k_vals[tab][(uint16_t)(code - hb->minc[len - 1] + hb->pos[len - 1])];
    const int run = sy >> 4, size = sy & 15;
    *dc = s->k == 0; *end = 0; *val = size != 0 || s->k == 0;
    uint32_t kn;
    if (s->k == 0) kn = 1;
    else if (run == 0 && size == 0) { *end = 1; kn = 64; }
    else { kn = s->k + run + 1; if (kn >= 64) *end = 1; }
    s->t += len + size;
    if (*end) { s->k = 0; s->j = s->j == 5 ? 0 : s->j + 1; } else s->k = kn;
    return len + size;
}
int main(int argc, char **argv)
{
    const uint32_t n = argc > 1 ? atoi(argv[1]) : 500, w = argc > 2 ? atoi(argv[2]) : 160, h = argc > 3 ? atoi(argv[3]) : 120;
    uint64_t cap = (uint64_t)n * w * h;
    uint8_t *blob = malloc(cap); uint64_t *offs = malloc(n * 8); uint32_t *lens = malloc(n * 4);
    for (int i = 0; i < 4; i++) build_bounds(&g_hb[i], k_bits[i]);
    amvo_synth_encode_batch(0xA11CE, 0, n, w, h, 0, blob, cap, offs, lens, 8);
    uint64_t nsym = 0, nbits = 0, hist[32] = {0}, ndc = 0, nend = 0;
    uint64_t steps[6][20] = {{0}};   /* policy x index bits */
    for (uint32_t f = 0; f < n; f++) {
        const uint8_t *c = blob + offs[f];
        uint32_t len = lens[f], nb = 0;
        uint8_t *bits = malloc((size_t)len * 8 + 64);
        for (uint32_t i = 2; i + 2 < len; i++) { uint8_t b = c[i]; for (int q = 7; q >= 0; q--) bits[nb++] = (b >> q) & 1; if (b == 0xff) i++; }
        /* the symbol sequence of the frame */
        static int L[200000], E[200000], D[200000];
        int m = 0; st3 s = {0, 0, 0};
        const uint32_t blocks = ((w + 15) / 16) * ((h + 15) / 16) * 6;
        uint32_t nblk = 0;
        while (nblk < blocks && s.t < nb) { int e, d, v; int l = sym(bits, nb, &s, &e, &d, &v); if (l < 0) break; L[m] = l; E[m] = e; D[m] = d; m++; nblk += e; nsym++; nbits += l; hist[l < 31 ? l : 31]++; ndc += d; nend += e; }
        for (int B = 9; B <= 13; B++) {
            /* policy 0: AC,AC pairs only (first must not end the block; second may).  policy 1: + DC,AC.  policy 2: any two
             * (block-ending first symbol too: needs the tables of the next context -- upper bound).  policy 3: triples, AC only */
            for (int pol = 0; pol < 4; pol++) {
                uint64_t st = 0;
                for (int i = 0; i < m;) {
                    int take = 1;
                    if (i + 1 < m && L[i] + L[i + 1] <= B) {
                        const int ok0 = !D[i] && !E[i] && !D[i + 1];
                        const int ok1 = !E[i] && !D[i + 1];
                        if ((pol == 0 || pol == 3) && ok0) take = 2;
                        if (pol == 1 && ok1) take = 2;
                        if (pol == 2) take = 2;
                        if (pol == 3 && take == 2 && i + 2 < m && !E[i + 1] && !D[i + 2] && L[i] + L[i + 1] + L[i + 2] <= B) take = 3;
                    }
                    i += take; st++;
                }
                steps[pol][B] += st;
            }
        }
        free(bits);
    }
    printf("%llu symbols, %.2f bits per symbol, %.1f%% DC, %.1f%% block ends\n", (unsigned long long)nsym, (double)nbits / nsym, 100.0 * ndc / nsym, 100.0 * nend / nsym);
    printf("bits per symbol histogram:"); for (int i = 1; i < 32; i++) if (hist[i]) printf(" %d:%.1f%%", i, 100.0 * hist[i] / nsym); printf("\n");
    for (int pol = 0; pol < 4; pol++) { printf("policy %d: steps per symbol by index bits", pol); for (int B = 9; B <= 13; B++) printf("  %d: %.3f", B, (double)steps[pol][B] / nsym); printf("\n"); }
    return 0;
}
