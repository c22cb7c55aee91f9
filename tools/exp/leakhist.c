// development: LD_PRELOAD malloc interposer -- histogram of LIVE heap blocks by requested size, dumped on demand (leakhist_dump(tag), found with dlsym by
// tests/cpp/mirror_threads_driver in its timing mode) and with a call-site sample for one size (LEAKHIST_TRACE_SIZE=n: backtrace of live blocks of n bytes).
//   gcc -O2 -shared -fPIC -o /tmp/leakhist.so tools/exp/leakhist.c -ldl -lpthread
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static void* (*real_malloc)(size_t);
static void (*real_free)(void*);
static void* (*real_calloc)(size_t, size_t);
static void* (*real_realloc)(void*, size_t);
static int (*real_memalign)(void**, size_t, size_t);
static void* (*real_aligned)(size_t, size_t);

#define NB (1u << 22)
typedef struct { void* p; size_t n; void* bt[6]; } ent;
static ent* tab;
static pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
static __thread int inside;
static size_t trace_size;
static char boot[65536]; static size_t boot_used;

static void init(void) {
    if (real_malloc) return;
    inside = 1;
    real_malloc = dlsym(RTLD_NEXT, "malloc"); real_free = dlsym(RTLD_NEXT, "free"); real_calloc = dlsym(RTLD_NEXT, "calloc");
    real_realloc = dlsym(RTLD_NEXT, "realloc"); real_memalign = dlsym(RTLD_NEXT, "posix_memalign"); real_aligned = dlsym(RTLD_NEXT, "aligned_alloc");
    tab = real_calloc(NB, sizeof(ent));
    const char* e = getenv("LEAKHIST_TRACE_SIZE"); trace_size = e ? (size_t)atol(e) : 0;
    inside = 0;
}
static void put(void* p, size_t n) {
    if (!p || inside || !tab) return;
    inside = 1;
    pthread_mutex_lock(&mu);
    size_t h = ((uintptr_t)p >> 4) * 0x9E3779B97F4A7C15ull >> 42;
    for (size_t k = 0; k < NB; k++) { ent* e = &tab[(h + k) & (NB - 1)]; if (!e->p || e->p == (void*)1) { e->p = p; e->n = n; if (trace_size && n == trace_size) backtrace(e->bt, 6); else e->bt[0] = 0; break; } }
    pthread_mutex_unlock(&mu);
    inside = 0;
}
static void del(void* p) {
    if (!p || inside || !tab) return;
    inside = 1;
    pthread_mutex_lock(&mu);
    size_t h = ((uintptr_t)p >> 4) * 0x9E3779B97F4A7C15ull >> 42;
    for (size_t k = 0; k < NB; k++) { ent* e = &tab[(h + k) & (NB - 1)]; if (!e->p) break; if (e->p == p) { e->p = (void*)1; break; } }
    pthread_mutex_unlock(&mu);
    inside = 0;
}
void* malloc(size_t n) { if (!real_malloc) { if (inside) { void* r = boot + boot_used; boot_used += (n + 15) & ~15ul; return r; } init(); } void* p = real_malloc(n); put(p, n); return p; }
void free(void* p) { if ((char*)p >= boot && (char*)p < boot + sizeof(boot)) return; if (!real_free) init(); del(p); real_free(p); }
void* calloc(size_t a, size_t b) { if (!real_calloc) { if (inside) { void* r = boot + boot_used; boot_used += (a * b + 15) & ~15ul; memset(r, 0, a * b); return r; } init(); } void* p = real_calloc(a, b); put(p, a * b); return p; }
void* realloc(void* q, size_t n) { if (!real_realloc) init(); if ((char*)q >= boot && (char*)q < boot + sizeof(boot)) { void* p = malloc(n); memcpy(p, q, n); return p; } del(q); void* p = real_realloc(q, n); put(p, n); return p; }
int posix_memalign(void** out, size_t al, size_t n) { if (!real_memalign) init(); int r = real_memalign(out, al, n); if (!r) put(*out, n); return r; }
void* aligned_alloc(size_t al, size_t n) { if (!real_aligned) init(); void* p = real_aligned(al, n); put(p, n); return p; }

typedef struct { size_t n, cnt; } hrow;
static int cmp(const void* a, const void* b) { const hrow* x = a; const hrow* y = b; size_t u = x->n * x->cnt, v = y->n * y->cnt; return u < v ? 1 : (u > v ? -1 : 0); }
void leakhist_dump(const char* tag) {
    inside = 1;
    pthread_mutex_lock(&mu);
    static hrow rows[1 << 16]; size_t nr = 0; size_t total = 0, blocks = 0;
    for (size_t k = 0; k < NB; k++) {
        ent* e = &tab[k]; if (!e->p || e->p == (void*)1) continue;
        total += e->n; blocks++;
        size_t r = 0; for (; r < nr; r++) if (rows[r].n == e->n) break;
        if (r == nr) { if (nr == (1 << 16)) continue; rows[nr].n = e->n; rows[nr].cnt = 0; nr++; }
        rows[r].cnt++;
    }
    qsort(rows, nr, sizeof(hrow), cmp);
    fprintf(stderr, "LEAKHIST %s: %zu live blocks, %.1f MB;", tag, blocks, total / 1048576.0);
    for (size_t r = 0; r < nr && r < 40; r++) fprintf(stderr, " %zux%zu", rows[r].n, rows[r].cnt);
    fprintf(stderr, "\n");
    if (trace_size) {
        int shown = 0;
        for (size_t k = 0; k < NB && shown < 6; k++) { ent* e = &tab[k]; if (!e->p || e->p == (void*)1 || e->n != trace_size || !e->bt[0]) continue;
            if ((k * 2654435761u) % 97 != 0) continue;
            char** sy = backtrace_symbols(e->bt, 6); fprintf(stderr, "LEAKHIST trace %zu:", e->n); for (int i = 0; i < 6 && sy; i++) fprintf(stderr, " | %s", sy[i]); fprintf(stderr, "\n"); shown++; }
    }
    pthread_mutex_unlock(&mu);
    inside = 0;
}
