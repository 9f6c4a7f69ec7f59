/*
 * c_abi_demo.c -- the C ABI of include/chq.h used from plain C, the way a Rust / Go / Java FFI binding would:
 * no Python, no torch, no C++.  It builds the record batches by hand as Arrow C Data Interface structs over malloc'ed
 * host buffers, rebuilds `value2 > 10.0` with the chq_expr_* constructors, and runs
 *
 *   1. chq_filter_record            one batch, host in / host out            (filter_task.rs:99)
 *   2. chq_filter_records           64 reference-sized batches, one launch    (the loop of filter_task.rs:78-126)
 *   3. chq_filter_records_coalesced the same, outputs joined
 *   4. chq_filter_project_record    filter + `id, value2 * 2.0 AS twice`      (materialize_files_task.rs:110)
 *
 * checking every result against a scalar loop in this file, and prints per-call timings of the C entry points
 * (PCIe staging both ways included).  Exit status 0 = everything matched.
 *
 *   make -C examples && ./examples/c_abi_demo [rows_per_batch] [batches]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../include/chq.h"

static double now_ms(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec / 1e6;
}

/* ---- a minimal Arrow C Data producer: struct<id:int32, value1:float32, value2:float32> over caller buffers ---- */
typedef struct {
  struct ArrowDeviceArray dev;      /* parent (struct) array */
  struct ArrowArray children[3];
  struct ArrowArray* child_ptrs[3];
  const void* parent_buffers[1];
  const void* child_buffers[3][2];
} HostBatch;

static void noop_release_array(struct ArrowArray* a) { a->release = NULL; }
static void noop_release_schema(struct ArrowSchema* s) { s->release = NULL; }

static void make_batch(HostBatch* b, int64_t n, const int32_t* id, const float* v1, const float* v2) {
  memset(b, 0, sizeof *b);
  const void* cols[3] = {id, v1, v2};
  for (int i = 0; i < 3; ++i) {
    struct ArrowArray* c = &b->children[i];
    b->child_buffers[i][0] = NULL;       /* no validity bitmap */
    b->child_buffers[i][1] = cols[i];
    c->length = n; c->null_count = 0; c->offset = 0; c->n_buffers = 2; c->buffers = b->child_buffers[i];
    c->release = noop_release_array;
    b->child_ptrs[i] = c;
  }
  b->parent_buffers[0] = NULL;
  b->dev.array.length = n; b->dev.array.n_buffers = 1; b->dev.array.buffers = b->parent_buffers;
  b->dev.array.n_children = 3; b->dev.array.children = b->child_ptrs;
  b->dev.array.release = noop_release_array;
  b->dev.device_id = -1; b->dev.device_type = ARROW_DEVICE_CPU;
}

static struct ArrowSchema g_schema, g_fields[3];
static struct ArrowSchema* g_field_ptrs[3];
static void make_schema(void) {
  static const char* names[3] = {"id", "value1", "value2"};
  static const char* fmts[3] = {"i", "f", "f"};
  memset(&g_schema, 0, sizeof g_schema);
  for (int i = 0; i < 3; ++i) {
    memset(&g_fields[i], 0, sizeof g_fields[i]);
    g_fields[i].format = fmts[i]; g_fields[i].name = names[i]; g_fields[i].release = noop_release_schema;
    g_field_ptrs[i] = &g_fields[i];
  }
  g_schema.format = "+s"; g_schema.name = ""; g_schema.n_children = 3; g_schema.children = g_field_ptrs;
  g_schema.release = noop_release_schema;
}

static void release_out(struct ArrowDeviceArray* a, struct ArrowSchema* s) {
  if (a->array.release) a->array.release(&a->array);
  if (s->release) s->release(s);
}

#define CHECK(call)                                                                     \
  do {                                                                                  \
    chq_status st_ = (call);                                                            \
    if (st_ != CHQ_OK) {                                                                \
      fprintf(stderr, "%s failed: %s: %s\n", #call, chq_status_name(st_), chq_ctx_last_error(ctx)); \
      return 1;                                                                         \
    }                                                                                   \
  } while (0)

/* Position-weighted digest of an output batch (arithmetic modulo 2^64): sum over columns c of W[c] * sum over rows k of
 * bits32(col_c[k]) * (k + 1).  The tests compare the digests this program prints with tests/golden/c_abi_demo_expected.json,
 * which scripts/make_c_abi_demo_fixture.py produced by running the ORACLE on the same generated data in the build container
 * (the scalar loop below stays as the in-program check). */
static uint64_t digest_columns(const struct ArrowArray* out, int ncols) {
  static const uint64_t W[3] = {1, 3, 7};
  uint64_t d = 0;
  for (int c = 0; c < ncols; ++c) {
    const uint32_t* v = (const uint32_t*)out->children[c]->buffers[1] + out->children[c]->offset;
    uint64_t s = 0;
    for (int64_t k = 0; k < out->length; ++k) s += (uint64_t)v[k] * (uint64_t)(k + 1);
    d += W[c] * s;
  }
  return d;
}

/* expected survivors of `value2 > 10.0` in rows [r0, r0 + n) */
static int64_t expect_rows(const float* v2, int64_t r0, int64_t n) {
  int64_t k = 0;
  for (int64_t i = 0; i < n; ++i) k += v2[r0 + i] > 10.0f;
  return k;
}
static int check_filtered(const struct ArrowArray* out, const int32_t* id, const float* v1, const float* v2, int64_t r0, int64_t n) {
  const int32_t* oid = (const int32_t*)out->children[0]->buffers[1] + out->children[0]->offset;
  const float* ov1 = (const float*)out->children[1]->buffers[1] + out->children[1]->offset;
  const float* ov2 = (const float*)out->children[2]->buffers[1] + out->children[2]->offset;
  int64_t k = 0;
  for (int64_t i = 0; i < n; ++i) {
    if (!(v2[r0 + i] > 10.0f)) continue;
    if (k >= out->length || oid[k] != id[r0 + i] || memcmp(&ov1[k], &v1[r0 + i], 4) || memcmp(&ov2[k], &v2[r0 + i], 4)) return 0;
    ++k;
  }
  return k == out->length;
}

int main(int argc, char** argv) {
  const int64_t rows = argc > 1 ? atoll(argv[1]) : 10000;   /* physical_planner.rs:323 */
  const int nb = argc > 2 ? atoi(argv[2]) : 64;
  const int64_t total = rows * nb;
  int32_t* id = malloc(total * 4);
  float* v1 = malloc(total * 4);
  float* v2 = malloc(total * 4);
  uint32_t x = 0xC0FFEEu;
  for (int64_t i = 0; i < total; ++i) {
    id[i] = (int32_t)i;
    x = x * 1664525u + 1013904223u; v1[i] = (float)(x >> 8) * (100.0f / 16777216.0f);
    x = x * 1664525u + 1013904223u; v2[i] = (float)(x >> 8) * (100.0f / 16777216.0f);
  }
  make_schema();
  HostBatch* batches = malloc(sizeof(HostBatch) * nb);
  const struct ArrowDeviceArray** ptrs = malloc(sizeof(void*) * nb);
  for (int b = 0; b < nb; ++b) { make_batch(&batches[b], rows, id + b * rows, v1 + b * rows, v2 + b * rows); ptrs[b] = &batches[b].dev; }

  chq_ctx* ctx = NULL;
  chq_status st = chq_ctx_create(0, NULL, &ctx);
  if (st != CHQ_OK) { fprintf(stderr, "chq_ctx_create: %s (no GPU?)\n", chq_status_name(st)); return 2; }

  /* value2 > 10.0  as  BinaryOp { Identifier("value2"), Gt, Value(Number("10.0", false)) } */
  chq_expr* pred = chq_expr_binary_op(chq_expr_identifier("value2"), CHQ_BINOP_GT, ">", chq_expr_number("10.0", 0));
  int failures = 0;
  uint64_t d_loop = 0, d_group = 0, d_join = 0, d_counts = 0, d_project = 0;
  int64_t n_loop = 0, n_group = 0, n_join = 0, n_project = 0;

  /* 1. one batch at a time: the reference's loop */
  double t0 = now_ms();
  for (int rep = 0; rep < 2; ++rep) {
    t0 = now_ms();
    for (int b = 0; b < nb; ++b) {
      struct ArrowDeviceArray out; struct ArrowSchema out_schema;
      CHECK(chq_filter_record(ctx, ptrs[b], &g_schema, NULL, pred, ARROW_DEVICE_CPU, &out, &out_schema));
      if (rep == 1 && !check_filtered(&out.array, id, v1, v2, b * rows, rows)) ++failures;
      if (rep == 1) { d_loop += (uint64_t)(b + 1) * digest_columns(&out.array, 3); n_loop += out.array.length; }
      release_out(&out, &out_schema);
    }
  }
  double t_loop = now_ms() - t0;

  /* 2. the whole queue in one call */
  struct ArrowDeviceArray* outs = calloc(nb, sizeof *outs);
  struct ArrowSchema* out_schemas = calloc(nb, sizeof *out_schemas);
  double t_group = 0;
  for (int rep = 0; rep < 2; ++rep) {
    t0 = now_ms();
    CHECK(chq_filter_records(ctx, nb, ptrs, &g_schema, NULL, pred, ARROW_DEVICE_CPU, outs, out_schemas));
    t_group = now_ms() - t0;
    for (int b = 0; b < nb; ++b) {
      if (rep == 1 && !check_filtered(&outs[b].array, id, v1, v2, b * rows, rows)) ++failures;
      if (rep == 1) { d_group += (uint64_t)(b + 1) * digest_columns(&outs[b].array, 3); n_group += outs[b].array.length; }
      release_out(&outs[b], &out_schemas[b]);
    }
  }
  chq_call_stats stats;
  chq_ctx_last_stats(ctx, &stats);

  /* 3. ... with the outputs joined */
  int64_t* per_record = malloc(sizeof(int64_t) * nb);
  double t_join = 0;
  for (int rep = 0; rep < 2; ++rep) {
    struct ArrowDeviceArray out; struct ArrowSchema out_schema;
    t0 = now_ms();
    CHECK(chq_filter_records_coalesced(ctx, nb, ptrs, &g_schema, NULL, pred, ARROW_DEVICE_CPU, &out, &out_schema, per_record));
    t_join = now_ms() - t0;
    if (rep == 1) {
      if (!check_filtered(&out.array, id, v1, v2, 0, total)) ++failures;
      for (int b = 0; b < nb; ++b) if (per_record[b] != expect_rows(v2, b * rows, rows)) ++failures;
      d_join = digest_columns(&out.array, 3); n_join = out.array.length;
      for (int b = 0; b < nb; ++b) d_counts += (uint64_t)(b + 1) * (uint64_t)per_record[b];
    }
    release_out(&out, &out_schema);
  }

  /* 4. filter -> project: SELECT id, value2 * 2.0 AS twice WHERE value2 > 10.0 */
  chq_expr* twice = chq_expr_binary_op(chq_expr_identifier("value2"), CHQ_BINOP_MULTIPLY, "*", chq_expr_number("2.0", 0));
  chq_expr* id_expr = chq_expr_identifier("id");
  chq_select_item items[2] = {{CHQ_ITEM_UNNAMED_EXPR, id_expr, NULL}, {CHQ_ITEM_EXPR_WITH_ALIAS, twice, "twice"}};
  {
    struct ArrowDeviceArray out; struct ArrowSchema out_schema;
    CHECK(chq_filter_project_record(ctx, pred, items, 2, ptrs[0], &g_schema, NULL, ARROW_DEVICE_CPU, &out, &out_schema));
    const int32_t* oid = (const int32_t*)out.array.children[0]->buffers[1] + out.array.children[0]->offset;
    const float* otw = (const float*)out.array.children[1]->buffers[1] + out.array.children[1]->offset;
    int64_t k = 0;
    for (int64_t i = 0; i < rows; ++i) {
      if (!(v2[i] > 10.0f)) continue;
      const float want = v2[i] * 2.0f;
      if (k >= out.array.length || oid[k] != id[i] || memcmp(&otw[k], &want, 4)) { ++failures; break; }
      ++k;
    }
    if (k != out.array.length || strcmp(out_schema.children[1]->name, "twice") != 0) ++failures;
    d_project = digest_columns(&out.array, 2); n_project = out.array.length;
    release_out(&out, &out_schema);
  }

  printf("c_abi_demo: %d batches x %lld rows (host buffers, PCIe both ways)\n", nb, (long long)rows);
  printf("  chq_filter_record per batch       : %8.3f ms  (%.1f us per call)\n", t_loop, t_loop * 1e3 / nb);
  printf("  chq_filter_records (one call)     : %8.3f ms  (%lld kernel launch%s)\n", t_group, (long long)stats.launches, stats.launches == 1 ? "" : "es");
  printf("  chq_filter_records_coalesced      : %8.3f ms\n", t_join);
  printf("  mismatches against the scalar loop: %d\n", failures);
  printf("digest filter_record rows=%lld sum=%016llx\n", (long long)n_loop, (unsigned long long)d_loop);
  printf("digest filter_records rows=%lld sum=%016llx\n", (long long)n_group, (unsigned long long)d_group);
  printf("digest filter_records_coalesced rows=%lld sum=%016llx counts=%016llx\n", (long long)n_join, (unsigned long long)d_join, (unsigned long long)d_counts);
  printf("digest filter_project_record rows=%lld sum=%016llx\n", (long long)n_project, (unsigned long long)d_project);

  chq_expr_free(pred); chq_expr_free(twice); chq_expr_free(id_expr);
  chq_ctx_destroy(ctx);
  free(batches); free(ptrs); free(outs); free(out_schemas); free(per_record); free(id); free(v1); free(v2);
  return failures ? 1 : 0;
}
