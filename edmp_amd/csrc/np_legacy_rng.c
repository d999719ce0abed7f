/* np_legacy_rng.c — NumPy's legacy RandomState.standard_normal stream, reproduced bit for bit, in parallel.
 *
 * Why: the reference draws all of its noise from the GLOBAL NumPy RandomState (diffusion/diffusion.py:126, 303:
 * np.random.multivariate_normal(0, I) == standard_normal draws), 91.75 M normals per 1024-trajectory scene.  NumPy makes
 * them one by one (~9 ns each, 0.85 s per scene on the GPU box) - twice the time the MI355X needs to denoise the
 * scene.  The algorithm (numpy/random/src/legacy/legacy-distributions.c: legacy_gauss; numpy/random/src/mt19937) is
 *
 *     has_gauss ? return the stored value :
 *     do { x1 = 2*d() - 1; x2 = 2*d() - 1; r2 = x1*x1 + x2*x2; } while (r2 >= 1.0 || r2 == 0.0);
 *     f = sqrt(-2*log(r2)/r2);  store f*x1;  return f*x2;            d() = ((a >> 5) * 67108864 + (b >> 6)) / 2^53
 *
 * with a, b two consecutive 32-bit MT19937 outputs.  Every ATTEMPT of the rejection loop consumes exactly four MT
 * words whether it is accepted or not, so attempt i always reads words [4i, 4i+4): the word stream is generated in
 * bulk (sequential, cheap), the attempts are evaluated independently across threads (the log / sqrt are the cost),
 * and the accepted pairs are compacted in order.  Same libm log/sqrt, no FMA contraction => identical doubles.
 *
 * Build: gcc -O3 -fPIC -shared -fopenmp -ffp-contract=off np_legacy_rng.c -o ../libedmp_nprng.so -lm
 * C ABI (ctypes binding in edmp_amd/nprng.py); host only, no GPU involved. */
#define _GNU_SOURCE
#include <math.h>
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MT_N 624
#define MT_M 397

typedef struct {
    uint32_t key[MT_N];
    int pos;
} mt_state;

/* regenerate the 624-word block (numpy/random/src/mt19937/mt19937.c: mt19937_gen).  Written as three loops over disjoint
 * source / destination ranges of at most 227 words (the recurrence x[i+624] = f(x[i], x[i+1], x[i+397]) reads nothing
 * closer than 227 words behind its own writes), so that the compiler vectorises them; integer-only: every ISA clone
 * gives the same words. */
#define MT_TWIST(u, v) ((((u) & 0x80000000u) | ((v) & 0x7fffffffu)) >> 1) ^ ((uint32_t)(-(int32_t)((v) & 1u)) & 0x9908b0dfu)
__attribute__((target_clones("avx2", "default"))) static void mt_gen(mt_state* s) {
    uint32_t* restrict k = s->key;
    uint32_t tmp[MT_N];
    int i;
    /* new[0..226] from old[0..227] and old[397..623] */
    for (i = 0; i < MT_N - MT_M; i++) tmp[i] = k[i + MT_M] ^ MT_TWIST(k[i], k[i + 1]);
    /* new[227..453] from old[227..454] and new[0..226] */
    for (i = MT_N - MT_M; i < 2 * (MT_N - MT_M); i++) tmp[i] = tmp[i - (MT_N - MT_M)] ^ MT_TWIST(k[i], k[i + 1]);
    /* new[454..622] from old[454..623] and new[227..395] */
    for (i = 2 * (MT_N - MT_M); i < MT_N - 1; i++) tmp[i] = tmp[i - (MT_N - MT_M)] ^ MT_TWIST(k[i], k[i + 1]);
    /* new[623] from old[623], NEW[0] and new[396] */
    tmp[MT_N - 1] = tmp[MT_M - 1] ^ MT_TWIST(k[MT_N - 1], tmp[0]);
    memcpy(k, tmp, sizeof(tmp));
    s->pos = 0;
}

/* the next `n` tempered outputs, in stream order */
__attribute__((target_clones("avx2", "default"))) static void mt_fill(mt_state* s, uint32_t* restrict out, int64_t n) {
    int64_t done = 0;
    while (done < n) {
        if (s->pos == MT_N) mt_gen(s);
        int64_t take = MT_N - s->pos;
        if (take > n - done) take = n - done;
        const uint32_t* restrict k = s->key + s->pos;
        uint32_t* restrict o = out + done;
        for (int64_t i = 0; i < take; i++) {
            uint32_t y = k[i];
            y ^= (y >> 11);
            y ^= (y << 7) & 0x9d2c5680u;
            y ^= (y << 15) & 0xefc60000u;
            y ^= (y >> 18);
            o[i] = y;
        }
        s->pos += (int)take;
        done += take;
    }
}

static inline double legacy_double(uint32_t a, uint32_t b) {
    return ((a >> 5) * 67108864.0 + (b >> 6)) / 9007199254740992.0;
}

/* one attempt: returns 1 and the pair (first returned value f*x2, then the stored f*x1) if accepted */
static inline int attempt(const uint32_t* w, double* v) {
    const double x1 = 2.0 * legacy_double(w[0], w[1]) - 1.0;
    const double x2 = 2.0 * legacy_double(w[2], w[3]) - 1.0;
    const double r2 = x1 * x1 + x2 * x2;
    if (r2 >= 1.0 || r2 == 0.0) return 0;
    const double f = sqrt(-2.0 * log(r2) / r2);
    v[0] = f * x2;
    v[1] = f * x1;
    return 1;
}

/* attempts per block (16 bytes of words + 16 bytes of candidate pairs each): 2^15 keeps the two word blocks and the
 * candidate pairs (1.5 MiB) inside the cache the team shares - measured on the GPU box (EPYC 9575F, 8 threads on one CCD):
 * 1.5 ns per normal at 2^15 against 2.2 at 2^18.  EDMP_NPRNG_BLOCK=<log2> overrides. */
static int64_t g_block_att = 0;
static pthread_once_t g_block_once = PTHREAD_ONCE_INIT;
static void block_att_init(void) {
    const char* e = getenv("EDMP_NPRNG_BLOCK");
    int lg = e ? atoi(e) : 15;
    if (lg < 10) lg = 10;
    if (lg > 22) lg = 22;
    g_block_att = (int64_t)1 << lg;
}
static int64_t block_att(void) { /* two contexts draw from two host threads: initialised exactly once */
    (void)pthread_once(&g_block_once, block_att_init);
    return g_block_att;
}
#define BLOCK_ATT (block_att())
#define MAX_THREADS 256

/* Thread placement.  The team hands 4 MiB/ms of words and candidate pairs from core to core: spread over a two-socket box by
 * the scheduler (the GPU boxes expose 256 CPUs and grant 16) the draws cost 3.0-3.9 ns per normal, confined to ONE last-level
 * cache domain 1.5 (scripts/nprng_bench.py).  At the first call the library picks the L3 domain of the CPU the caller is
 * running on (sysfs cache/index3/shared_cpu_list, intersected with the process affinity mask); every team thread restricts
 * itself to that domain (the caller's own mask is restored when the call returns) and the team is capped at the domain's
 * physical cores.  EDMP_NPRNG_PIN=0 disables, EDMP_NPRNG_CPUS=a-b[,c-d] names the CPUs explicitly. */
static cpu_set_t g_domain;
static int g_domain_state = 0; /* 0 unknown, 1 pinned, -1 no pinning */
static int g_domain_cores = 0;

static int parse_cpu_list(const char* txt, cpu_set_t* set) {
    CPU_ZERO(set);
    int n = 0;
    const char* p = txt;
    while (*p) {
        char* end;
        long a = strtol(p, &end, 10);
        if (end == p) break;
        long b = a;
        p = end;
        if (*p == '-') {
            b = strtol(p + 1, &end, 10);
            p = end;
        }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++) {
            CPU_SET((int)c, set);
            n++;
        }
        if (*p == ',') p++;
        else break;
    }
    return n;
}

static pthread_once_t g_domain_once = PTHREAD_ONCE_INIT;
static void pick_domain_once(void) {
    g_domain_state = -1;
    const char* pin = getenv("EDMP_NPRNG_PIN");
    if (pin && atoi(pin) == 0) return;
    cpu_set_t allowed, dom;
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return;
    const char* list = getenv("EDMP_NPRNG_CPUS");
    char buf[4096] = {0};
    if (!list) {
        int cpu = sched_getcpu();
        if (cpu < 0) return;
        char path[128];
        snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", cpu);
        FILE* f = fopen(path, "r");
        if (!f) return;
        size_t got = fread(buf, 1, sizeof(buf) - 1, f);
        fclose(f);
        if (!got) return;
        list = buf;
    }
    if (parse_cpu_list(list, &dom) < 2) return;
    CPU_AND(&g_domain, &dom, &allowed);
    const int n = CPU_COUNT(&g_domain);
    if (n < 2) return;
    /* physical cores: count CPUs whose first SMT sibling (thread_siblings_list) is themselves */
    int cores = 0;
    for (int c = 0; c < CPU_SETSIZE; c++) {
        if (!CPU_ISSET(c, &g_domain)) continue;
        char path[128], sib[256] = {0};
        snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", c);
        FILE* f = fopen(path, "r");
        int first = c;
        if (f) {
            if (fread(sib, 1, sizeof(sib) - 1, f)) first = atoi(sib);
            fclose(f);
        }
        if (first == c || !CPU_ISSET(first, &g_domain)) cores++;
    }
    g_domain_cores = cores > 0 ? cores : n;
    g_domain_state = 1;
}
/* every caller sees the finished decision (0 is never observable after this returns): with a plain lazy static a second
 * draw thread could read the intermediate -1 and run unpinned */
static void pick_domain(void) { (void)pthread_once(&g_domain_once, pick_domain_once); }


/* number of threads a draw will actually use for `requested`, and the size of the cache domain (0: not pinned) */
int edmp_nprng_team(int requested, int* domain_cores) {
    pick_domain();
    if (domain_cores) *domain_cores = g_domain_state == 1 ? g_domain_cores : 0;
    if (g_domain_state == 1 && requested > g_domain_cores) return g_domain_cores;
    return requested;
}

/* sequential, exact-length generation of out[o..n) (used for the tail): evaluates a speculative batch of attempts and
 * rewinds the MT state to the last attempt NumPy would have consumed */
static void tail_exact(mt_state* st, int* has_gauss, double* gauss, double* out, int64_t o, int64_t n, uint32_t* words, double* cand,
                       unsigned char* ok) {
    while (o < n) {
        const int64_t need_pairs = (n - o + 1) / 2;
        int64_t att = (int64_t)((double)need_pairs * 1.28) + 16; /* acceptance rate pi/4 */
        if (att > BLOCK_ATT) att = BLOCK_ATT;
        mt_state before = *st;
        mt_fill(st, words, att * 4);
        for (int64_t i = 0; i < att; i++) ok[i] = (unsigned char)attempt(words + 4 * i, cand + 2 * i);
        int64_t used = 0;
        for (int64_t i = 0; i < att && o < n; i++) {
            used = i + 1;
            if (!ok[i]) continue;
            out[o++] = cand[2 * i];
            if (o < n) {
                out[o++] = cand[2 * i + 1];
            } else { /* the second value of the pair stays cached in the RandomState, like NumPy */
                *has_gauss = 1;
                *gauss = cand[2 * i + 1];
            }
        }
        if (used < att) { /* rewind: replay exactly the words of the attempts that were consumed */
            *st = before;
            int64_t skip = used * 4;
            while (skip > 0) {
                if (st->pos == MT_N) mt_gen(st);
                int64_t take = MT_N - st->pos;
                if (take > skip) take = skip;
                st->pos += (int)take;
                skip -= take;
            }
        }
    }
}

/* Fill out[0..n) with the values np.random.standard_normal(n) would return from the RandomState whose MT19937 state
 * is (key, *pos, *has_gauss, *gauss); the state is advanced exactly as NumPy would advance it.  Returns 0, or -1 on
 * allocation failure (state and output untouched). */
int edmp_nprng_standard_normal(uint32_t* key, int* pos, int* has_gauss, double* gauss, double* out, int64_t n, int nthreads) {
    if (n <= 0) return 0;
    uint32_t* words = (uint32_t*)malloc((size_t)BLOCK_ATT * 4 * sizeof(uint32_t));
    double* cand = (double*)malloc((size_t)BLOCK_ATT * 2 * sizeof(double));
    unsigned char* ok = (unsigned char*)malloc((size_t)BLOCK_ATT);
    if (!words || !cand || !ok) {
        free(words);
        free(cand);
        free(ok);
        return -1;
    }
    mt_state st;
    memcpy(st.key, key, sizeof(st.key));
    st.pos = *pos;
    int64_t o = 0;
    if (*has_gauss) {
        out[o++] = *gauss;
        *has_gauss = 0;
        *gauss = 0.0;
    }
#ifdef _OPENMP
    if (nthreads < 1) nthreads = omp_get_max_threads();
    if (nthreads > MAX_THREADS) nthreads = MAX_THREADS;
    nthreads = edmp_nprng_team(nthreads, NULL);
#else
    nthreads = 1;
#endif
    cpu_set_t caller_mask;
    const int pinned = (g_domain_state == 1) && sched_getaffinity(0, sizeof(caller_mask), &caller_mask) == 0;
    /* bulk phase: while more values are still needed than one block can possibly yield (2 per attempt), every attempt
     * of the block is consumed, so no length bookkeeping is needed.  One parallel region: thread 0 produces the word
     * block b+1 (MT19937 is sequential) while the other threads evaluate block b - attempts + in-place compaction per
     * thread chunk, then an ordered copy-out.  The block produced speculatively when the loop ends is un-done by
     * restoring the MT state saved before it. */
    uint32_t* words2 = (uint32_t*)malloc((size_t)BLOCK_ATT * 4 * sizeof(uint32_t));
    if (!words2) {
        free(words);
        free(cand);
        free(ok);
        return -1;
    }
    if (n - o > 2 * (int64_t)BLOCK_ATT) {
        uint32_t* wbuf[2] = {words, words2};
        mt_state saved = st;  /* state before the block sitting in wbuf[cur ^ 1] */
        int64_t cnt[MAX_THREADS + 1];
        int cont = 1, nt_used = 1;
        mt_fill(&st, wbuf[0], (int64_t)BLOCK_ATT * 4);
#pragma omp parallel num_threads(nthreads)
        {
#ifdef _OPENMP
            const int t = omp_get_thread_num(), nt = omp_get_num_threads();
#else
            const int t = 0, nt = 1;
#endif
            /* this thread only, and only for this call: libgomp keeps the team of a master thread alive, and other OpenMP
             * work started from the same thread (torch, NumPy) would otherwise inherit workers confined to one L3 domain */
            cpu_set_t my_mask;
            const int my_pinned = pinned && sched_getaffinity(0, sizeof(my_mask), &my_mask) == 0;
            if (my_pinned) (void)sched_setaffinity(0, sizeof(g_domain), &g_domain);
            /* consumers: all threads when alone, otherwise threads 1..nt-1 */
            const int nc = nt > 1 ? nt - 1 : 1, c = nt > 1 ? t - 1 : 0;
            if (t == 0) nt_used = nt;
            for (int blk = 0;; ++blk) {
                const uint32_t* w4 = wbuf[blk & 1];
                if (nt > 1 && t == 0) { /* producer: next block, speculatively */
                    saved = st;
                    mt_fill(&st, wbuf[(blk & 1) ^ 1], (int64_t)BLOCK_ATT * 4);
                }
                int64_t a0 = 0, mine = 0;
                if (c >= 0) {
                    a0 = (int64_t)BLOCK_ATT * c / nc;
                    const int64_t a1 = (int64_t)BLOCK_ATT * (c + 1) / nc;
                    int64_t w = a0;
                    for (int64_t i = a0; i < a1; i++) {
                        double v[2];
                        if (attempt(w4 + 4 * i, v)) { /* w <= i: compaction inside the chunk's own slots */
                            cand[2 * w] = v[0];
                            cand[2 * w + 1] = v[1];
                            ++w;
                        }
                    }
                    mine = w - a0;
                    cnt[c] = mine;
                }
#pragma omp barrier
                if (c >= 0) {
                    int64_t off = 0;
                    for (int q = 0; q < c; q++) off += cnt[q];
                    memcpy(out + o + 2 * off, cand + 2 * a0, (size_t)mine * 2 * sizeof(double));
                }
#pragma omp barrier
#pragma omp single
                {
                    int64_t tot = 0;
                    for (int q = 0; q < nc; q++) tot += cnt[q];
                    o += 2 * tot;
                    cont = (n - o > 2 * (int64_t)BLOCK_ATT);
                    if (nt == 1 && cont) { /* no producer thread: make the next block now */
                        saved = st;
                        mt_fill(&st, wbuf[(blk & 1) ^ 1], (int64_t)BLOCK_ATT * 4);
                    }
                } /* implicit barrier */
                if (!cont) break;
            }
            if (my_pinned) (void)sched_setaffinity(0, sizeof(my_mask), &my_mask);
        }
        if (nt_used > 1) st = saved; /* drop the speculative block */
    }
    free(words2);
    if (pinned) (void)sched_setaffinity(0, sizeof(caller_mask), &caller_mask); /* (the master restored its mask inside the region; kept as a safety net) */
    tail_exact(&st, has_gauss, gauss, out, o, n, words, cand, ok);
    free(words);
    free(cand);
    free(ok);
    memcpy(key, st.key, sizeof(st.key));
    *pos = st.pos;
    return 0;
}

int edmp_nprng_version(void) { return 2; }
