/* fake_hip.c -- TEST INFRASTRUCTURE: a CPU stand-in for the device side of include/sayuri_hip.h, so that the host
 * side of the hot path (HipForwardPipe: staging, leaf-batch collector, ring pump, hand-out of results;
 * reference src/neural/batch_forward_pipe.cc:7-193) can be exercised -- and run under ThreadSanitizer -- on a box
 * without a GPU.  Loaded with RTLD_GLOBAL ahead of libsayuri_hip.so by tests/test_collector_cpu.py; never part of
 * the product.
 *
 * The "network" is a fixed, cheap function of each sample's own planes (fake_eval below), so a test can tell for
 * every reply whether it belongs to the request that asked for it.  submit() is asynchronous like the real one:
 * a worker thread per context finishes a batch FAKE_HIP_DELAY_US microseconds after it was enqueued, two tickets
 * may be in flight.
 *
 * Latency model for the multi-rank readiness runs (tools/fake8.py): with FAKE_HIP_SERIAL_US=T the device is SERIAL like
 * the real one -- a batch occupies it for T microseconds whatever its size (3 800 us = one 256-batch of the 20b x 256
 * network on an MI355X), the next one starts when the previous has finished -- and FAKE_HIP_CHEAP=1 replaces the
 * per-plane checksum network by one that costs the host next to nothing (policy = the first bit planes plus a hash of
 * the position), so that the host cores measured are the engine's, not the stand-in's. */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../../include/sayuri_hip.h"

struct job {
    int n, state; /* 0 free, 1 queued, 2 done */
    const float* planes;
    const unsigned* records; /* cheap mode: the packed records themselves */
    int binary;
    float* owned; /* planes expanded from packed records (freed when the job is done) */
    const int* bsz;
    float *prob, *pass, *misc, *own;
    struct timespec due;
};
struct sayuri_hip_ctx {
    sayuri_hip_netdesc desc;
    int board, max_batch, next, stop;
    long delay_us, serial_us;
    int cheap;
    struct timespec busy_until; /* serial model: when the device is free again */
    struct job jobs[2];
    int order[2], n_order; /* FIFO of queued tickets */
    pthread_mutex_t mu;
    pthread_cond_t cv_work, cv_done;
    pthread_t worker;
    long submits, evals;
};

static void fake_eval(const sayuri_hip_ctx* c, int n, const float* planes, const int* bsz, float* prob, float* pass,
                      float* misc, float* own) {
    const int B2 = c->board * c->board, C = c->desc.input_channels;
    const int PC = c->desc.probabilities_channels, PP = c->desc.pass_probability_outputs, VM = c->desc.value_misc_outputs;
    for (int i = 0; i < n; ++i) {
        const float* x = planes + (size_t)i * C * B2;
        double s = 0;
        for (int k = 0; k < C * B2; ++k) s += x[k] * (double)(1 + k % 7);
        const int bs = bsz ? bsz[i] : c->board;
        for (int k = 0; k < PC; ++k)
            for (int p = 0; p < B2; ++p) prob[((size_t)i * PC + k) * B2 + p] = x[(size_t)k * B2 + p] + 0.5f * x[(size_t)5 * B2 + p] + (float)k;
        for (int k = 0; k < PP; ++k) pass[(size_t)i * PP + k] = (float)(s * 0.001) + (float)k;
        for (int k = 0; k < VM; ++k) misc[(size_t)i * VM + k] = (float)(s * 0.002) - (float)k + (float)bs;
        for (int p = 0; p < B2; ++p) own[(size_t)i * B2 + p] = x[(size_t)7 * B2 + p] - x[(size_t)8 * B2 + p];
    }
}

/* FAKE_HIP_CHEAP: a few operations per output value.  From planes: policy k = plane k; from packed records: policy k =
 * bit plane k plus a per-position hash in [0, 1) (different positions prefer different moves). */
static void cheap_eval(const sayuri_hip_ctx* c, int n, const float* planes, const unsigned* rec, int binary, const int* bsz,
                       float* prob, float* pass, float* misc, float* own) {
    const int B = c->board, B2 = B * B, C = c->desc.input_channels;
    const int PC = c->desc.probabilities_channels, PP = c->desc.pass_probability_outputs, VM = c->desc.value_misc_outputs;
    const int words = binary * 12 + 8;
    memset(prob, 0, sizeof(float) * (size_t)n * PC * B2);
    memset(own, 0, sizeof(float) * (size_t)n * B2);
    for (int i = 0; i < n; ++i) {
        const int bs = bsz ? bsz[i] : B;
        unsigned h = 2166136261u;
        if (rec) for (int k = 0; k < 24; ++k) h = (h ^ rec[(size_t)i * words + k]) * 16777619u;
        for (int k = 0; k < PC; ++k)
            for (int y = 0; y < bs; ++y)
                for (int x = 0; x < bs; ++x) {
                    float v;
                    if (rec) {
                        const int cell = y * bs + x;
                        unsigned g = (h + (unsigned)cell * 2654435761u) * 2246822519u;
                        v = (float)((rec[(size_t)i * words + k * 12 + (cell >> 5)] >> (cell & 31)) & 1u) + (float)(g >> 8) * (1.0f / 16777216.0f);
                    } else {
                        v = planes[((size_t)i * C + k) * B2 + y * B + x];
                    }
                    prob[((size_t)i * PC + k) * B2 + y * B + x] = v;
                }
        for (int k = 0; k < PP; ++k) pass[(size_t)i * PP + k] = -2.0f;
        for (int k = 0; k < VM; ++k) misc[(size_t)i * VM + k] = (float)((h >> (k & 15)) & 255u) * (1.0f / 256.0f) - 0.5f;
    }
}

/* packed records (csrc/host/packed_planes.h) -> NN-grid fp32 planes, what the device kernel pack_bits_kernel does */
static float* expand_records(const sayuri_hip_ctx* c, int n, const unsigned* rec, int binary, const int* bsz) {
    const int B = c->board, B2 = B * B, C = c->desc.input_channels, words = binary * 12 + 8;
    float* out = (float*)calloc((size_t)n * C * B2, sizeof(float));
    for (int i = 0; i < n; ++i) {
        const unsigned* r = rec + (size_t)i * words;
        const int bs = bsz ? bsz[i] : B;
        for (int ch = 0; ch < C; ++ch)
            for (int y = 0; y < bs; ++y)
                for (int x = 0; x < bs; ++x) {
                    const int cell = y * bs + x;
                    float v;
                    if (ch < binary) v = (float)((r[ch * 12 + (cell >> 5)] >> (cell & 31)) & 1u);
                    else memcpy(&v, &r[binary * 12 + (ch - binary)], sizeof v);
                    out[((size_t)i * C + ch) * B2 + y * B + x] = v;
                }
    }
    return out;
}

static void* worker_main(void* arg) {
    sayuri_hip_ctx* c = (sayuri_hip_ctx*)arg;
    pthread_mutex_lock(&c->mu);
    for (;;) {
        while (!c->stop && c->n_order == 0) pthread_cond_wait(&c->cv_work, &c->mu);
        if (c->stop && c->n_order == 0) break;
        const int t = c->order[0];
        struct job* j = &c->jobs[t];
        const struct timespec due = j->due;
        pthread_mutex_unlock(&c->mu);
        clock_nanosleep(CLOCK_MONOTONIC, TIMER_ABSTIME, &due, NULL);
        if (c->cheap) cheap_eval(c, j->n, j->planes, j->records, j->binary, j->bsz, j->prob, j->pass, j->misc, j->own);
        else fake_eval(c, j->n, j->planes, j->bsz, j->prob, j->pass, j->misc, j->own);
        free(j->owned);
        j->owned = NULL;
        pthread_mutex_lock(&c->mu);
        j->state = 2;
        c->order[0] = c->order[1];
        --c->n_order;
        pthread_cond_broadcast(&c->cv_done);
    }
    pthread_mutex_unlock(&c->mu);
    return NULL;
}

int sayuri_hip_device_count(void) {
    const char* e = getenv("FAKE_HIP_DEVICES");
    return e ? atoi(e) : 1;
}
const char* sayuri_hip_last_error(void) { return "fake_hip: no error text"; }

sayuri_hip_ctx* sayuri_hip_create(int device, const sayuri_hip_netdesc* desc, int max_batch, int board, int use_fp16) {
    (void)device; (void)use_fp16;
    sayuri_hip_ctx* c = (sayuri_hip_ctx*)calloc(1, sizeof(*c));
    c->desc = *desc;
    c->desc.blocks = NULL;
    c->board = board;
    c->max_batch = max_batch;
    const char* d = getenv("FAKE_HIP_DELAY_US");
    c->delay_us = d ? atol(d) : 200;
    const char* su = getenv("FAKE_HIP_SERIAL_US");
    c->serial_us = su ? atol(su) : 0;
    c->cheap = getenv("FAKE_HIP_CHEAP") != NULL;
    pthread_mutex_init(&c->mu, NULL);
    pthread_cond_init(&c->cv_work, NULL);
    pthread_cond_init(&c->cv_done, NULL);
    pthread_create(&c->worker, NULL, worker_main, c);
    return c;
}
void sayuri_hip_destroy(sayuri_hip_ctx* c) {
    if (!c) return;
    pthread_mutex_lock(&c->mu);
    c->stop = 1;
    pthread_cond_broadcast(&c->cv_work);
    pthread_mutex_unlock(&c->mu);
    pthread_join(c->worker, NULL);
    free(c);
}
int sayuri_hip_load_tensor(sayuri_hip_ctx* c, int layer_id, int kind, const float* host, size_t n) {
    (void)c; (void)layer_id; (void)kind;
    return host && n > 0 ? 0 : -1;
}
void* sayuri_hip_host_alloc(size_t bytes) { return calloc(1, bytes ? bytes : 1); }
void sayuri_hip_host_free(void* p) { free(p); }

int sayuri_hip_forward(sayuri_hip_ctx* c, int n, const float* planes, const int* bsz, float* prob, float* pass,
                       float* misc, float* own) {
    if (!c || n <= 0 || n > c->max_batch) return -1;
    fake_eval(c, n, planes, bsz, prob, pass, misc, own);
    return 0;
}
static int submit_any(sayuri_hip_ctx* c, int n, const float* planes, float* owned, const int* bsz, float* prob, float* pass,
                      float* misc, float* own, int* ticket);

int sayuri_hip_submit(sayuri_hip_ctx* c, int n, const float* planes, const int* bsz, float* prob, float* pass,
                      float* misc, float* own, int* ticket) {
    return submit_any(c, n, planes, NULL, bsz, prob, pass, misc, own, ticket);
}
int sayuri_hip_forward_packed(sayuri_hip_ctx* c, int n, const unsigned* records, int binary, const int* bsz, float* prob,
                              float* pass, float* misc, float* own) {
    if (!c || !records || n <= 0 || n > c->max_batch) return -1;
    float* x = expand_records(c, n, records, binary, bsz);
    fake_eval(c, n, x, bsz, prob, pass, misc, own);
    free(x);
    return 0;
}
int sayuri_hip_submit_packed(sayuri_hip_ctx* c, int n, const unsigned* records, int binary, const int* bsz, float* prob,
                             float* pass, float* misc, float* own, int* ticket) {
    if (!c || !records || n <= 0 || n > c->max_batch) return -1;
    if (c->cheap) {
        pthread_mutex_lock(&c->mu);
        struct job* j = &c->jobs[c->next];
        j->records = records;
        j->binary = binary;
        pthread_mutex_unlock(&c->mu);
        return submit_any(c, n, NULL, NULL, bsz, prob, pass, misc, own, ticket);
    }
    float* x = expand_records(c, n, records, binary, bsz);
    const int rc = submit_any(c, n, x, x, bsz, prob, pass, misc, own, ticket);
    if (rc) free(x);
    return rc;
}
static int submit_any(sayuri_hip_ctx* c, int n, const float* planes, float* owned, const int* bsz, float* prob, float* pass,
                      float* misc, float* own, int* ticket) {
    if (!c || n <= 0 || n > c->max_batch) return -1;
    pthread_mutex_lock(&c->mu);
    const int t = c->next;
    struct job* j = &c->jobs[t];
    if (j->state != 0) { pthread_mutex_unlock(&c->mu); return -1; } /* more than two batches in flight */
    c->next ^= 1;
    j->n = n; j->planes = planes; j->owned = owned; j->bsz = bsz; j->prob = prob; j->pass = pass; j->misc = misc; j->own = own;
    if (planes) j->records = NULL;
    clock_gettime(CLOCK_MONOTONIC, &j->due);
    long add_us = c->delay_us;
    if (c->serial_us > 0) { /* serial device: starts when the previous batch has finished */
        if (c->busy_until.tv_sec > j->due.tv_sec || (c->busy_until.tv_sec == j->due.tv_sec && c->busy_until.tv_nsec > j->due.tv_nsec))
            j->due = c->busy_until;
        add_us = c->serial_us;
    }
    j->due.tv_nsec += (add_us % 1000000) * 1000;
    j->due.tv_sec += add_us / 1000000 + j->due.tv_nsec / 1000000000;
    j->due.tv_nsec %= 1000000000;
    c->busy_until = j->due;
    j->state = 1;
    c->order[c->n_order++] = t;
    ++c->submits;
    c->evals += n;
    *ticket = t;
    pthread_cond_signal(&c->cv_work);
    pthread_mutex_unlock(&c->mu);
    return 0;
}
int sayuri_hip_wait(sayuri_hip_ctx* c, int t) {
    if (!c || t < 0 || t > 1) return -1;
    pthread_mutex_lock(&c->mu);
    if (c->jobs[t].state == 0) { pthread_mutex_unlock(&c->mu); return -1; }
    while (c->jobs[t].state != 2) pthread_cond_wait(&c->cv_done, &c->mu);
    c->jobs[t].state = 0;
    pthread_mutex_unlock(&c->mu);
    return 0;
}
int sayuri_hip_query(sayuri_hip_ctx* c, int t) {
    if (!c || t < 0 || t > 1) return -1;
    pthread_mutex_lock(&c->mu);
    const int s = c->jobs[t].state;
    pthread_mutex_unlock(&c->mu);
    return s == 0 ? -1 : s == 2;
}
size_t sayuri_hip_device_bytes(const sayuri_hip_ctx* c) { (void)c; return 0; }
int sayuri_hip_last_chains(const sayuri_hip_ctx* c) { (void)c; return 1; }
int sayuri_hip_tower_state(const sayuri_hip_ctx* c) { (void)c; return 0; }

/* ---- the resident-input entry points bench.py times (upload / run / sync / download / time_runs + the per-class event
 * bookkeeping): the serial model again -- a run occupies the "device" for FAKE_HIP_SERIAL_US (or FAKE_HIP_DELAY_US) --
 * so that bench.py's whole control flow, including its multi-rank path, can be executed without a GPU
 * (tests/test_dropin_cpu.py::test_bench_two_ranks_on_the_fake_device). */
static int g_up_n = 0;
static char g_mark[48];
static int g_mark_launches = 0;
static double g_mark_ms = 0;
static void fake_busy(const sayuri_hip_ctx* c) {
    const long us = c->serial_us > 0 ? c->serial_us : c->delay_us;
    struct timespec ts = {us / 1000000, (us % 1000000) * 1000};
    nanosleep(&ts, NULL);
}
int sayuri_hip_upload(sayuri_hip_ctx* c, int n, const float* planes, const int* bsz) {
    (void)planes; (void)bsz;
    if (!c || n <= 0 || n > c->max_batch) return -1;
    g_up_n = n;
    return 0;
}
int sayuri_hip_run(sayuri_hip_ctx* c) { if (!c || g_up_n == 0) return -1; fake_busy(c); return 0; }
int sayuri_hip_sync(sayuri_hip_ctx* c) { return c ? 0 : -1; }
int sayuri_hip_download(sayuri_hip_ctx* c, float* prob, float* pass, float* misc, float* own) {
    (void)prob; (void)pass; (void)misc; (void)own;
    return c && g_up_n ? 0 : -1;
}
int sayuri_hip_time_runs(sayuri_hip_ctx* c, int iters, float* total_ms) {
    if (!c || g_up_n == 0 || iters < 0) return -1;
    struct timespec a, b;
    clock_gettime(CLOCK_MONOTONIC, &a);
    for (int i = 0; i < iters; ++i) fake_busy(c);
    clock_gettime(CLOCK_MONOTONIC, &b);
    const double ms = (b.tv_sec - a.tv_sec) * 1e3 + (b.tv_nsec - a.tv_nsec) * 1e-6;
    if (total_ms) *total_ms = (float)ms;
    g_mark_launches = 0;
    g_mark_ms = 0;
    if (!strcmp(g_mark, "tower_run")) { g_mark_launches = iters; g_mark_ms = ms; }
    return 0;
}
int sayuri_hip_mark_kernel(sayuri_hip_ctx* c, const char* name) {
    (void)c;
    strncpy(g_mark, name ? name : "", sizeof(g_mark) - 1);
    return 0;
}
int sayuri_hip_timed_stat(sayuri_hip_ctx* c, sayuri_hip_kernel_stat* row) {
    if (!c || !row) return -1;
    memset(row, 0, sizeof(*row));
    strncpy(row->name, g_mark, sizeof(row->name) - 1);
    row->launches = g_mark_launches;
    row->total_ms = (float)g_mark_ms;
    row->flops = 4.379e12 * g_mark_launches; /* the 20b x 256 tower on 256 boards */
    row->bytes = 4.84e9 * g_mark_launches;
    return 0;
}
int sayuri_hip_profile_run(sayuri_hip_ctx* c, sayuri_hip_kernel_stat* rows, int cap) { (void)c; (void)rows; (void)cap; return 0; }
