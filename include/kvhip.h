/* kvhip.h — C ABI of the MI355X-native KvVariable hot path (libkvhip.so).
 *
 * This is the drop-in boundary: every entry point below is what a TensorFlow custom-op
 * shim (or any FFI: ctypes, cgo, JNI) binds in place of one reference OpKernel.  Plain
 * pointers and sizes only — no TF, torch or C++ types.  All `ids`, `counts`, `grad`,
 * `out`, `keys`, `values` … buffers are DEVICE pointers on the table's GPU (HBM
 * resident); scalars are passed by value.  Every call enqueues its kernels on `stream`
 * (a hipStream_t, NULL = the default stream) and returns without synchronising unless
 * its comment says "synchronous".
 *
 * Return value: 0 (KV_OK) or a tensorflow::error::Code-compatible integer, the same
 * category the reference's OP_REQUIRES would have set; kv_last_error() gives the text
 * for the calling thread.  Citations are relative to the reference tree
 * (intelligent-machine-learning/tfplus).
 */
#ifndef KVHIP_H_
#define KVHIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct kv_table* kv_handle_t; /* opaque; one KvVariable<K, float> resource */
typedef void* kv_stream_t;            /* hipStream_t */

/* status codes (tensorflow/core/protobuf/error_codes.proto numbering) */
#define KV_OK 0
#define KV_INVALID_ARGUMENT 3    /* errors::InvalidArgument */
#define KV_RESOURCE_EXHAUSTED 8  /* HBM allocation failed */
#define KV_FAILED_PRECONDITION 9 /* "Failed to use uninitialized variables" */
#define KV_UNIMPLEMENTED 12
#define KV_INTERNAL 13 /* HIP runtime error */

/* key_dtype / value_dtype (tensorflow DataType numbering, kernels/kv_variable_ops.cc:149-156) */
#define KV_DT_FLOAT 1
#define KV_DT_INT32 3
#define KV_DT_INT64 9
#define KV_DT_UINT64 23

/* kv_scatter_update ops (kernels/kv_variable_interface.h:44-52) */
#define KV_SCATTER_ASSIGN 0
#define KV_SCATTER_ADD 1
#define KV_SCATTER_SUB 2
#define KV_SCATTER_MUL 3
#define KV_SCATTER_DIV 4
#define KV_SCATTER_MIN 5
#define KV_SCATTER_MAX 6

const char* kv_last_error(void);

/* ---- lifecycle ------------------------------------------------------------------------- */

/* Replaces CreateKvVariableOp::Compute (kernels/kv_variable_ops.cc:58-116) + KvVariable ctor
 * (kernels/kv_variable.h:92-114).  `dim` = value_shape.num_elements(); `enter_threshold` is
 * clamped to uint16 like SaturateMaxFrequency.  `capacity_hint` pre-sizes the HBM hash index
 * and row slab for that many keys (0 = small default; both grow on demand).  Keys of every
 * supported key_dtype are handled as their 64-bit two's-complement pattern; `ids` buffers are
 * int64 (KV_DT_INT64 / KV_DT_UINT64) or int32 (KV_DT_INT32). */
int kv_create(int key_dtype, int value_dtype, int dim, int enter_threshold, int64_t capacity_hint,
              int device, kv_handle_t* out);
/* Replaces DestroyKvVariableOp (kernels/kv_variable_ops.cc:295-323). Synchronous. */
int kv_destroy(kv_handle_t h);
/* Grows index + slab so that `capacity` keys fit without a rehash. Synchronous. */
int kv_reserve(kv_handle_t h, int64_t capacity);

/* Replaces InitKvVariableOp (kernels/kv_variable_ops.cc:188-200) -> InitRandomValues
 * (kernels/kv_variable.h:184-206): stores the [rows, dim] fp32 init table; first call wins. */
int kv_init_table(kv_handle_t h, const float* table, int64_t rows, kv_stream_t stream);
/* KvVariableIsInitializedV2 (kernels/kv_variable_ops.cc:202-220) */
int kv_is_initialized(kv_handle_t h, int* out);

/* Test / reproducibility hooks (no reference counterpart): the reference stamps rows with
 * time(NULL)/86400 (kernels/utility.cc:38-40) and picks init rows with std::rand()
 * (kernels/kv_variable.h:889-898).  day < 0 restores the wall clock. */
int kv_set_clock_days(kv_handle_t h, int day);
int kv_set_seed(kv_handle_t h, uint64_t seed);

/* ---- readback / statistics (synchronous: they return host scalars) ---------------------- */

/* KvVariableSizeV2 -> size() (kernels/kv_variable.h:139-156): rows that are not blacklisted
 * and whose frequency >= enter_threshold. */
int kv_size(kv_handle_t h, int64_t* out, kv_stream_t stream);
/* KvVariableShapeV2 dim 0 (kernels/kv_variable.h:177-182): every key in the map. */
int kv_map_size(kv_handle_t h, int64_t* out, kv_stream_t stream);
/* KvVariableFrequency -> sum_freq() (kernels/kv_variable.h:158-175) */
int kv_sum_freq(kv_handle_t h, int64_t* out, kv_stream_t stream);
/* Per-key meta for tests: freq_words[i] = (day << 16) | frequency, flags[i] bit0 = blacklist,
 * bit1 = under_threshold, bit7 = key present (EmbeddingValue, kernels/embedding_value.h:225-235). */
int kv_get_meta(kv_handle_t h, const int64_t* ids, int64_t n, uint32_t* freq_words, uint8_t* flags,
                kv_stream_t stream);

/* ---- lookup (HOT LOOP 1) ------------------------------------------------------------------ */

/* Replaces KvVariableGatherOrInsertOp / ...WithCountsOp::Compute
 * (kernels/kv_variable_ops.cc:498-538, 564-606) -> KvVariable::FindOrInsert
 * (kernels/kv_variable.h:263-380).  out[i, :] = row(ids[i]); missing keys are inserted with
 * 0.5*(T[r1]+T[r2]); every occurrence bumps the saturating uint16 frequency by 1 or by
 * min(counts[i], 65535) and stamps the day; blacklisted keys read as zeros.  `counts` may be NULL.
 * n == 0 is a no-op.  Fails with KV_FAILED_PRECONDITION if the init table is not set and a key
 * would have to be inserted (the reference dereferences an empty tensor there).  This entry point keeps nothing
 * for a later apply (and is ~12 us faster per 1 M ids for it): a training step uses kv_gather_or_insert_tok. */
int kv_gather_or_insert(kv_handle_t h, const void* ids, const int32_t* counts, int64_t n,
                        float* out, kv_stream_t stream);
/* kv_gather_or_insert with the ids and their occurrence counts interleaved as int64 pairs
 * (id, count) [n][2] — the payload kv_bucket_by_owner builds and the all-to-all delivers, used
 * as is by the owner of a sharded table (counts saturate at 65535 like WithCounts). */
int kv_gather_or_insert_pairs(kv_handle_t h, const int64_t* id_count_pairs, int64_t n, float* out,
                              kv_stream_t stream);

/* The same lookup, additionally naming the batch: *token (never 0 on success for n <= 2^21) identifies the
 * index of `ids` that the lookup leaves in the table's workspace — the input positions sorted by key, each
 * key's row.  A kv_apply_*_tok call on the same table with the same ids, n and token takes that index over
 * instead of rebuilding it (a training step applies the ids it has just looked up: the TF shim threads the
 * token from the forward op to the optimizer op).  Any other batch op on the table in between simply makes
 * the token stale; a stale or zero token means "build the index", never a wrong result.  The caller
 * guarantees only that `ids` holds the same values at both calls. */
typedef uint64_t kv_batch_token_t;
int kv_gather_or_insert_tok(kv_handle_t h, const void* ids, const int32_t* counts, int64_t n,
                            float* out, kv_batch_token_t* token, kv_stream_t stream);

/* Replaces KvVariableGatherOrZerosOp::Compute (kernels/kv_variable_ops.cc:348-405) -> FindOrZeros
 * (kernels/kv_variable.h:239-254): no insert, no frequency change, misses read as zeros. */
int kv_gather_or_zeros(kv_handle_t h, const void* ids, int64_t n, float* out, kv_stream_t stream);

/* ---- sparse optimizer apply (HOT LOOP 2) ---------------------------------------------------
 * Common contract: `ids` [n] and `grad` [n, dim] as the op receives them.  Repeated ids are
 * first combined the TF-core way (tf.unique + tf.unsorted_segment_sum, what
 * optimizer._resource_apply_sparse_duplicate_indices does in front of the reference op,
 * python/ops/variable_scope.py:1096-1106) inside the same call — the "segment-reduced
 * scatter-add fused with the row update".  Rows whose var frequency < enter_threshold are
 * skipped; new keys are inserted with the init rule; slot tables count one hit per apply. */

/* Replaces KvVariableGroupSparseApplyAdamV4Op::Compute (kernels/training_ops.cc:6988-7213,
 * version = 4) and ...V3Op (:5709-5965, version = 3).  `m_v_linear` has dim 3*dim ([m|v|z]).
 * Scalars in the op's input order (ops/training_ops.cc:1266-1285): lr, beta1_power,
 * beta2_power, beta1 ("beat1"), beta2, epsilon, l1, l2, l21. */
int kv_apply_group_adam(kv_handle_t var, kv_handle_t m_v_linear, const float* grad, const void* ids,
                        int64_t n, float lr, float beta1_power, float beta2_power, float beta1,
                        float beta2, float epsilon, float l1, float l2, float l21, int version,
                        kv_stream_t stream);
/* Replaces KvVariableSparseApplyAdagradOp::Compute (kernels/training_ops.cc:1372-1498);
 * argument order of ops/training_ops.cc:214-226. */
int kv_apply_adagrad(kv_handle_t var, kv_handle_t accum, float lr, const float* grad,
                     const void* ids, int64_t n, int update_slots, kv_stream_t stream);
/* Replaces KvVariableSparseGroupSparseApplyFtrlOp<..., has_l2_shrinkage = true>::Compute
 * (kernels/training_ops.cc:532-778); argument order of ops/training_ops.cc:135-150. */
int kv_apply_sparse_group_ftrl(kv_handle_t var, kv_handle_t accum, kv_handle_t linear,
                               const float* grad, const void* ids, int64_t n, float lr, float l1,
                               float l2, float l21, float l2_shrinkage, float lr_power,
                               kv_stream_t stream);

/* The same three ops at the reference's REAL op boundary.  In an unchanged TF1 graph the reference's processor patch
 * (python/ops/variable_scope.py:1096-1106) sends a KvVariable's IndexedSlices gradient through TF-core's
 * _deduplicate_indexed_slices, so KvVariableGroupSparseApplyAdamV4 / KvVariableSparseApplyAdagrad /
 * KvVariableSparseGroupSparseApplyFtrlV2 receive indices that are already UNIQUE and gradient rows that are already
 * summed (kernels/training_ops.cc:7011-7021).  The caller PROMISES that `ids` holds no id twice; the op is then one
 * launch — one lane group per id: index probe (a key the table does not hold is inserted: FindOrInsertUnsafe,
 * kv_variable.h:382-416), then gradient row, var row, slot row and both records in one round trip, the fused row
 * update — instead of the batch pipeline's tile dedup + tile sums + partition pass, whose answers the promise makes
 * known.  Same arguments, checks and results as the plain ops (rows, slots, frequency words, flags: the same bits for
 * the same summed gradient).  A BROKEN promise is detected on the device, not raced silently: every row record carries
 * the serial of the last unique-apply launch that touched it, flipped by one returning atomic per id; an id listed
 * twice raises the table's error word and the NEXT call on the table returns KV_INVALID_ARGUMENT (that batch was not
 * applied as the reference applies repeated ids — sequentially).  Embedding dims the kernel does not serve (not a
 * multiple of 4, or above 256) take the batch pipeline, which needs no promise — and so does a call made under
 * stream capture (the launch serial lives on the host: a replayed launch would meet its own stamps).  No batch token
 * is involved; a pending lookup pass of the table is settled first. */
int kv_apply_group_adam_unique(kv_handle_t var, kv_handle_t m_v_linear, const float* grad, const void* ids, int64_t n,
                               float lr, float beta1_power, float beta2_power, float beta1, float beta2, float epsilon,
                               float l1, float l2, float l21, int version, kv_stream_t stream);
int kv_apply_adagrad_unique(kv_handle_t var, kv_handle_t accum, float lr, const float* grad, const void* ids,
                            int64_t n, int update_slots, kv_stream_t stream);
int kv_apply_sparse_group_ftrl_unique(kv_handle_t var, kv_handle_t accum, kv_handle_t linear, const float* grad,
                                      const void* ids, int64_t n, float lr, float l1, float l2, float l21,
                                      float l2_shrinkage, float lr_power, kv_stream_t stream);

/* The three optimizer ops with the batch token of the lookup that preceded them (0 = none). */
int kv_apply_group_adam_tok(kv_handle_t var, kv_handle_t m_v_linear, const float* grad, const void* ids,
                            int64_t n, float lr, float beta1_power, float beta2_power, float beta1,
                            float beta2, float epsilon, float l1, float l2, float l21, int version,
                            kv_batch_token_t token, kv_stream_t stream);
int kv_apply_adagrad_tok(kv_handle_t var, kv_handle_t accum, float lr, const float* grad, const void* ids,
                         int64_t n, int update_slots, kv_batch_token_t token, kv_stream_t stream);
int kv_apply_sparse_group_ftrl_tok(kv_handle_t var, kv_handle_t accum, kv_handle_t linear, const float* grad,
                                   const void* ids, int64_t n, float lr, float l1, float l2, float l21,
                                   float l2_shrinkage, float lr_power, kv_batch_token_t token,
                                   kv_stream_t stream);

/* Optimizer slot tables are KvVariables of their own (python/ops/variable_scope.py:1027-1088, created with
 * colocate_with(var), training/group_adam.py:142), probed with the var's keys on every apply.  Here a var's
 * index entry also remembers the key's row in ONE slot table (the first slot of the optimizer: m_v_linear /
 * accum), so the apply reaches the slot row without a second probe; a remembered row is checked against the
 * slot row's own key before use.  The first apply that pairs a var with a slot table attaches it; this call
 * does so explicitly and, in one pass over the var's rows, fills in the rows of the keys both tables
 * already hold (after a checkpoint restore, or a table filled by kv_insert / kv_gather_or_insert). */
int kv_attach_slot(kv_handle_t var, kv_handle_t slot, kv_stream_t stream);

/* Reduction order of the gradient rows of an id that is repeated in a batch (the optimizer ops, kv_dedup_segment_sum, the
 * scatter family).  on = 0 (the default): as they arrive.  on = 1, deterministic: an order fixed by the input positions
 * alone — in-tile ranks by position instead of LDS-atomic arrival, a key's tiles in tile order — so the same batch gives
 * bit-identical optimizer state on every run.  on = 2, occurrence order: one by one in input order starting from +0, the sum
 * TF-core's unsorted_segment_sum takes on the CPU in front of the reference's optimizer ops
 * (_deduplicate_indexed_slices; python/ops/variable_scope.py:1096-1106) — bit for bit, so a batch with repeated ids leaves
 * the state the reference's CPU path leaves to the same 1e-6 as a batch of unique ids.  A key's sum is then ONE chain of
 * additions: the table's ops take the sorted-position pipeline (2^21 ids per call) and a key that fills much of the batch
 * is summed by one wave (DESIGN.md section 3b has the cost).  Implies on = 1.  Single-table ops only: KV_UNIMPLEMENTED from the
 * kv_multi_* ops and kv_shard_create for such a table (and from this call for a table a kv_shard serves) — the sharded
 * ops add the senders' partial sums in rank order, which no CPU run defines.  KV_INVALID_ARGUMENT for other values. */
#define KV_ORDER_ARRIVAL 0
#define KV_ORDER_FIXED 1
#define KV_ORDER_OCCURRENCE 2
int kv_set_deterministic(kv_handle_t h, int on);

/* Row math of the optimizer ops applied to var table `h` (training_ops.cc:7166-7195 and the other per-id bodies).
 * on = 0 (the default): IEEE-754 sqrt and division sequences, bit-for-bit what the reference's Eigen arithmetic gives
 * from the same summed gradient.  on = 1: every sqrt and division of the update is ONE hardware instruction
 * (v_sqrt_f32 / v_rcp_f32 / v_rsq_f32, 1 ulp each).  Measured at configs[1]: -1.3 us of the 64 us apply kernel (the
 * kernel is bound by its memory round trips, not by its arithmetic), and an element whose update cancels (|x| three
 * orders below its row's scale) can leave the parity tests' rtol 1e-6 / atol 1e-9 (one element in 39 000 did): hence
 * opt-in.  SparseGroupFtrl is the optimizer it suits least: its linear term (sqrt(n) - sqrt(a)) / lr * x cancels by
 * construction, so each root's ulp reaches the state multiplied by |x| / lr (tests/test_gpu_fast_math.py states the
 * bound: 4e-6 absolute at lr = 0.05 against 1e-7 for the other three).  A table in deterministic mode always uses the
 * IEEE sequences. */
int kv_set_fast_math(kv_handle_t h, int on);

/* Counters of the table's own ops, for tests and tools (no kernel reads them).  KV_STAT_MIRROR_APPLIES: optimizer applies
 * with `h` as the var that worked on the rows' slot mirrors (the lean update: the slot row's frequency word and flags are
 * read and written in the var row's own record line, csrc/kv_device.h SlotMirror); KV_STAT_MIRROR_EPOCHS: how often the
 * mirrors of `h` (as the var) were flushed back and invalidated because another op entered one of the two tables. */
#define KV_STAT_MIRROR_APPLIES 0
#define KV_STAT_MIRROR_EPOCHS 1
int kv_get_stat(kv_handle_t h, int which, int64_t* value);

/* Brings the host's upper bounds of the table's row count up to date (one synchronisation): a lookup or apply that
 * follows can then take `max_new_ids` more ids without consulting the device — what a stream capture needs, where a
 * synchronisation is not allowed.  KV_RESOURCE_EXHAUSTED when the table would have to grow for that many ids (it
 * grows here, outside the capture, if it can).  The table's batch workspace is sized by the calls themselves: issue the
 * op once with the batch length outside the capture; a captured call that would have to grow it returns
 * KV_FAILED_PRECONDITION before it queues anything (the capture stays valid). */
int kv_prepare_capture(kv_handle_t h, int64_t max_new_ids, kv_stream_t stream);

/* The TF-core step on its own (for callers that want the [U, dim] IndexedSlices):
 * uniq_ids [n], summed [n, dim] are filled for the first *num_unique entries; inverse [n]
 * (may be NULL) maps each input position to its unique row.  Unique order is unspecified
 * (TF-core gives first-occurrence order; consumers here are order-independent).  `h` only
 * supplies the device, stream workspace and dim.  Synchronous (returns the host count). */
int kv_dedup_segment_sum(kv_handle_t h, const void* ids, const float* grad, int64_t n,
                         int64_t* uniq_ids, float* summed, int32_t* inverse, int64_t* num_unique,
                         kv_stream_t stream);

/* ---- export / import ("next" rows) -------------------------------------------
 * Replaces ReadKvVariableOp / KvVariableExport (kernels/kv_variable_ops.cc:325-346, 779-860) ->
 * ExportValues (kernels/dynamic_save.hpp:47-195).  Two-phase: kv_export_count fills
 * counts[3] = {num_rows, blacklist_nums, freq_nums} on the host (synchronous), then the caller
 * allocates and kv_export_fill writes keys [num_rows], values [num_rows, dim],
 * blacklist [blacklist_nums], freq_keys / freq_values [freq_nums] (any of the last three may be
 * NULL when its count is 0).  first_n: 2 = keys+values; >3 adds blacklist; >4 adds the uint32
 * frequency words.  Row order is unspecified (the reference's is hash-map iteration order).
 * The fill must follow its count directly: any other op on the table in between (the buffers would no longer be
 * known to fit) makes it return KV_FAILED_PRECONDITION; count again.  The same holds for kv_export_delta_count /
 * _fill and for kv_delete_with_timestamp's dry run / real run. */
int kv_export_count(kv_handle_t h, int first_n, int64_t* counts, kv_stream_t stream);
int kv_export_fill(kv_handle_t h, int first_n, int64_t* keys, float* values, int64_t* blacklist,
                   int64_t* freq_keys, uint32_t* freq_values, kv_stream_t stream);
/* Delta lists.  Replaces the SUPPORT_DELTA_EXPORT / SUPPORT_PREDICTION_DELTA_EXPORT environment
 * switches read by the KvVariable constructor (kernels/kv_variable.h:100-111): while on, training
 * lookups, scatters / inserts, optimizer applies (keys the update reaches, on the var and every slot
 * table) and deletes remember their keys (kv_variable.h:316,451,685,747,772,791-799).  Off by default. */
int kv_set_delta_tracking(kv_handle_t h, int support_delta_export, int support_prediction_delta_export);
/* Replaces KvVariableFullOrDeltaExport with need_full_export = false (kernels/kv_variable_ops.cc:
 * 1064-1095) -> DeltaExport (kernels/dynamic_save.hpp:198-451).  Two-phase like kv_export_*:
 * counts[4] = {num_rows, blacklist_nums, freq_nums, delete_nums}; then keys [num_rows], values
 * [num_rows, dim], blacklist, freq_keys / freq_values (uint32 words; 0 for deleted keys), delete_keys.
 * first_n <= 3 (prediction export) reads train + prediction lists and moves blacklisted keys to
 * delete_keys; first_n > 4 adds the frequency words of every listed key.  The fill ends the export:
 * the lists are emptied / handed on as dynamic_save.hpp:432-443 does (kv_export_fill with first_n > 2
 * does the same, :179-192; kv_import empties both, dynamic_restore.hpp:258-259). */
int kv_export_delta_count(kv_handle_t h, int first_n, int64_t* counts, kv_stream_t stream);
int kv_export_delta_fill(kv_handle_t h, int first_n, int64_t* keys, float* values, int64_t* blacklist,
                         int64_t* freq_keys, uint32_t* freq_values, int64_t* delete_keys,
                         kv_stream_t stream);
/* Replaces KvVariableImport (kernels/kv_variable_ops.cc:862-940) -> ImportValues
 * (kernels/dynamic_restore.hpp:29-195): clears the table, then loads keys/values, the
 * blacklist and the frequency words (NULL / 0 to skip). */
int kv_import(kv_handle_t h, const int64_t* keys, const float* values, int64_t n,
              const int64_t* blacklist, int64_t n_blacklist, const int64_t* freq_keys,
              const uint32_t* freq_values, int64_t n_freq, kv_stream_t stream);

/* Replaces the delta branch of KvVariableFullOrDeltaImport[V2] (ops/kv_variable_ops.cc:576-631,
 * kernels/kv_variable_ops.cc:854-939 -> DeltaImport kernels/dynamic_restore.hpp:29-155): the table
 * is NOT cleared; keys are inserted or overwritten (their blacklist mark is lifted and
 * under_threshold re-evaluated), blacklist keys are marked (first_n > 3) or removed (first_n <= 3,
 * the inference load mode), frequency words are set on keys that exist, delete_keys are removed.
 * need_full_import == true is kv_import. */
int kv_import_delta(kv_handle_t h, const int64_t* keys, const float* values, int64_t n,
                    const int64_t* blacklist, int64_t n_blacklist, const int64_t* freq_keys,
                    const uint32_t* freq_values, int64_t n_freq, const int64_t* delete_keys,
                    int64_t n_delete, int first_n, kv_stream_t stream);

/* Replaces KvVariableInsertV2 (kernels/kv_variable_ops.cc:703-747) -> InsertOrUpdate
 * (kernels/kv_variable.h:423-485): row(ids[i]) = values[i, :] (insert or overwrite). */
int kv_insert(kv_handle_t h, const void* ids, const float* values, int64_t n, kv_stream_t stream);
/* Replaces KvVariableScatter{Update,Add,Sub,Mul,Div,Min,Max}V2 (kernels/kv_variable_ops.cc:
 * 1097-1161) -> ScatterUpdate (kernels/kv_variable.h:616-734): row = row <op> updates[i]; missing
 * keys are inserted with the init rule first; blacklisted rows are left untouched.  Repeated
 * ids: add / sub apply the SUM of their update rows (the reference applies each occurrence in
 * turn), mul / div their PRODUCT, min / max their minimum / maximum — every occurrence counts, as in the
 * reference (this step is synchronous); plain update keeps one of the occurrences (the reference's result
 * depends on its thread interleaving there). */
int kv_scatter_update(kv_handle_t h, const void* ids, const float* updates, int64_t n, int op,
                      kv_stream_t stream);

/* tf.unique_with_counts on the GPU (what embedding_lookup_sparse runs before the lookup,
 * python/ops/embedding_ops.py:362-372, and what the sharded path runs before the exchange):
 * uniq [n] / uniq_counts [n] (may be NULL; per-occurrence `counts` or 1 each, summed, saturating at
 * 65535 like the frequency they feed) are filled for the first *num_unique entries, inverse [n]
 * (may be NULL) maps input positions to them.  Order unspecified.  num_unique (host) makes the
 * call synchronous; pass NULL and num_unique_dev (device, int64) to leave the count on the device
 * and keep the stream running (kv_bucket_by_owner takes it as n_dev). */
int kv_unique(kv_handle_t h, const void* ids, const int32_t* counts, int64_t n, int64_t* uniq,
              int32_t* uniq_counts, int32_t* inverse, int64_t* num_unique, int64_t* num_unique_dev,
              kv_stream_t stream);

/* ---- table hygiene ------------------------------------------------------
 * KvVariableGetCountV2 (ops/kv_variable_ops.cc:349-358 -> KvVariable::GetCount kv_variable.h:503-524):
 * counts[i] = frequency (low 16 bits) of ids[i], 0 when absent.
 * KvVariableGetTimeStamp (ops :688-697 -> GetTimeStamp kv_variable.h:526-561): days[i] = day stamp
 * (high 16 bits) of ids[i], today's day number when absent.  Both asynchronous. */
int kv_get_count(kv_handle_t h, const void* ids, int64_t n, int32_t* counts, kv_stream_t stream);
int kv_get_timestamp(kv_handle_t h, const void* ids, int64_t n, uint32_t* days, kv_stream_t stream);
/* KvVariableDelete (ops :681-685 -> KvVariable::Delete kv_variable.h:737-755 -> DeleteKey
 * table_manager.h:405-416): the keys disappear from the table (absent keys are ignored); their
 * rows are recycled by later inserts.  *num_deleted (may be NULL) = keys actually removed.
 * KvVariableDeleteWithTimestamp (ops :699-707 -> kv_variable.h:757-789): removes every key whose
 * day stamp is > 0 and at least (uint16) threshold days before today.  dry_run != 0 only counts
 * (*count); dry_run == 0 removes and writes the removed keys to delete_keys [>= that count].
 * Both synchronous. */
int kv_delete(kv_handle_t h, const void* ids, int64_t n, int64_t* num_deleted, kv_stream_t stream);
int kv_delete_with_timestamp(kv_handle_t h, int threshold, int dry_run, int64_t* delete_keys,
                             int64_t* count, kv_stream_t stream);

/* BatchKvVariableGatherOrZerosV2 (ops/kv_variable_ops.cc:297-308, kernels/kv_variable_ops.cc:431-470):
 * outs[i] [ns[i], dim_i] = GatherOrZeros(tables[i], ids[i] [ns[i]]) for i < num_tables, all tables
 * on one device, dims free to differ — one kernel launch for the whole batch of tables (the
 * reference loops).  Asynchronous. */
int kv_batch_gather_or_zeros(int num_tables, const kv_handle_t* tables, const void* const* ids,
                             const int64_t* ns, float* const* outs, kv_stream_t stream);

/* ---- many tables, one launch per pipeline stage (new: the reference runs one op per table; a CTR
 * step has tens of small lookups / applies, e.g. example/dcn/train.py:219-300 with 26 features) ---
 * kv_multi_gather_or_insert == kv_gather_or_insert(tables[i], ids[i], counts ? counts[i] : NULL,
 * ns[i], outs[i]) for every i, and kv_multi_apply_group_adam == kv_apply_group_adam(vars[i],
 * slots[i], grads[i], ids[i], ns[i], <shared scalars>, version) for every i, but with 3 resp. 2
 * kernel launches in total (grid.y = table).  All tables of one call: same device, same dim, same
 * key dtype, listed once; dims must be multiples of 4 for the optimizer.  Asynchronous. */
int kv_multi_gather_or_insert(int num_tables, const kv_handle_t* tables, const void* const* ids,
                              const int32_t* const* counts, const int64_t* ns, float* const* outs,
                              kv_stream_t stream);
int kv_multi_apply_group_adam(int num_tables, const kv_handle_t* vars, const kv_handle_t* slots,
                              const float* const* grads, const void* const* ids, const int64_t* ns,
                              float lr, float beta1_power, float beta2_power, float beta1, float beta2,
                              float epsilon, float l1, float l2, float l21, int version,
                              kv_stream_t stream);
/* the same for kv_apply_adagrad and kv_apply_sparse_group_ftrl (config 5 mixes GroupAdam and
 * SparseGroupFtrl tables: one call per (optimizer, dim) group) */
int kv_multi_apply_adagrad(int num_tables, const kv_handle_t* vars, const kv_handle_t* accums, float lr,
                           const float* const* grads, const void* const* ids, const int64_t* ns,
                           int update_slots, kv_stream_t stream);
int kv_multi_apply_sparse_group_ftrl(int num_tables, const kv_handle_t* vars, const kv_handle_t* accums,
                                     const kv_handle_t* linears, const float* const* grads,
                                     const void* const* ids, const int64_t* ns, float lr, float l1,
                                     float l2, float l21, float l2_shrinkage, float lr_power,
                                     kv_stream_t stream);
/* The batch token (see kv_gather_or_insert_tok) for the batched ops: tokens[i] names the index the batched lookup left
 * in tables[i]'s workspace; the batched optimizer ops given the same ids and those tokens skip their index pass when
 * EVERY token is still valid (else all tables are indexed again — stale tokens are always safe).  tokens == NULL:
 * the plain ops. */
int kv_multi_gather_or_insert_tok(int num_tables, const kv_handle_t* tables, const void* const* ids,
                                  const int32_t* const* counts, const int64_t* ns, float* const* outs,
                                  kv_batch_token_t* tokens, kv_stream_t stream);
int kv_multi_apply_group_adam_tok(int num_tables, const kv_handle_t* vars, const kv_handle_t* slots,
                                  const float* const* grads, const void* const* ids, const int64_t* ns, float lr,
                                  float beta1_power, float beta2_power, float beta1, float beta2, float epsilon, float l1,
                                  float l2, float l21, int version, const kv_batch_token_t* tokens, kv_stream_t stream);
int kv_multi_apply_adagrad_tok(int num_tables, const kv_handle_t* vars, const kv_handle_t* accums, float lr,
                               const float* const* grads, const void* const* ids, const int64_t* ns,
                               int update_slots, const kv_batch_token_t* tokens, kv_stream_t stream);
int kv_multi_apply_sparse_group_ftrl_tok(int num_tables, const kv_handle_t* vars, const kv_handle_t* accums,
                                         const kv_handle_t* linears, const float* const* grads,
                                         const void* const* ids, const int64_t* ns, float lr, float l1, float l2,
                                         float l21, float l2_shrinkage, float lr_power,
                                         const kv_batch_token_t* tokens, kv_stream_t stream);
/* ... and with the caller's promise that no table's ids hold an id twice (kv_apply_*_unique above: what the ops of an
 * unchanged TF graph receive): ONE launch for all tables (grid.y = table), the promise guarded per table as there. */
int kv_multi_apply_group_adam_unique(int num_tables, const kv_handle_t* vars, const kv_handle_t* m_v_linears,
                                     const float* const* grads, const void* const* ids, const int64_t* ns, float lr,
                                     float beta1_power, float beta2_power, float beta1, float beta2, float epsilon,
                                     float l1, float l2, float l21, int version, kv_stream_t stream);
int kv_multi_apply_adagrad_unique(int num_tables, const kv_handle_t* vars, const kv_handle_t* accums, float lr,
                                  const float* const* grads, const void* const* ids, const int64_t* ns,
                                  int update_slots, kv_stream_t stream);
int kv_multi_apply_sparse_group_ftrl_unique(int num_tables, const kv_handle_t* vars, const kv_handle_t* accums,
                                            const kv_handle_t* linears, const float* const* grads,
                                            const void* const* ids, const int64_t* ns, float lr, float l1, float l2,
                                            float l21, float l2_shrinkage, float lr_power, kv_stream_t stream);


/* embedding_lookup_sparse on a KvVariable (python/ops/embedding_ops.py:279-441), fused: the
 * reference runs unique_with_counts -> GatherOrInsert[WithCounts] -> gather(idx) -> (x weights) ->
 * segment_sum / sparse_segment_{sum,mean,sqrt_n}.  ids [n] are sp_ids.values, segment_ids [n] are
 * sp_ids.indices[:, 0] (ascending, int32 or int64 = segment_dtype), weights [n] = sp_weights.values
 * or NULL.  out [num_segments, dim]; a segment without ids is a zero row (weighted: 0/0 as in the
 * reference).  Missing keys are inserted; count_occurrences != 0 adds every occurrence of a key to
 * its frequency (the WithCounts path taken when enter_threshold > 0, embedding_ops.py:373-382),
 * 0 adds 1 per distinct key.  n <= 2^21 per call. */
#define KV_COMBINER_SUM 0
#define KV_COMBINER_MEAN 1
#define KV_COMBINER_SQRTN 2
int kv_lookup_sparse(kv_handle_t h, const void* ids, const void* segment_ids, int segment_dtype,
                     const float* weights, int64_t n, int64_t num_segments, int combiner,
                     int count_occurrences, float* out, kv_stream_t stream);

/* tf.unsorted_segment_sum(data [n, dim], segment_ids [n] int32, num_segments) on the batch pipeline
 * (no float atomics: a segment named by 100 000 rows costs the same as 100 000 segments): out
 * [num_segments, dim]; segments nobody names are zero rows, ids outside [0, num_segments) are
 * dropped.  `h` supplies device, dim and workspace.  Used by the sharded backward pass to sum
 * gradients in the order the forward pass already exchanged.  Asynchronous.  n <= 2^21. */
int kv_unsorted_segment_sum(kv_handle_t h, const int32_t* segment_ids, const float* data, int64_t n,
                            int64_t num_segments, float* out, kv_stream_t stream);

/* ---- multi-GPU: tables sharded by id over the GPUs of a node (new design; the reference has no
 * communication layer — its only sharding rule is `ids % num_shards`, python/ops/embedding_ops.py:121-127 and
 * kernels/utility.h:90-107 ModKeyImpl) ------------------------------------------------------------------------
 * Ownership rules of an id: */
#define KV_OWNER_HASH 0 /* (mix64(id) >> 32) % world: balanced whatever the ids look like (the default); the high half: the index homes keys by the low bits */
#define KV_OWNER_MOD 1  /* floor_mod(id, world): the reference's rule, for checkpoints partitioned by it */

/* Counting sort of `ids` [n] by owner rank: out_ids [n] holds the ids grouped by owner (rank 0's first),
 * perm[j] = input position of out_ids[j], counts_dev[world] (device, int64) = ids per owner — the send counts of a
 * variable-size all-to-all.  n_dev (device, may be NULL): the real length min(n, *n_dev) when it is still on the
 * device.  Optional extras (NULL to skip): pairs_out [n][2] int64 = (id, id_counts[i] or 1) in bucket order;
 * pos_out [n] int32 = position of input i in the bucket order (inverse of perm).  `h` supplies device, key dtype
 * and scratch only. */
int kv_bucket_by_owner(kv_handle_t h, const void* ids, int64_t n, const int64_t* n_dev, int world, int owner_rule,
                       int64_t* out_ids, int32_t* perm, int64_t* counts_dev, const int32_t* id_counts,
                       int64_t* pairs_out, int32_t* pos_out, kv_stream_t stream);

/* A communicator over the GPUs of one node: RCCL (grouped ncclSend / ncclRecv: xGMI is point to point, one pair per
 * link) on a stream of its own.  id128 = the 128-byte unique id kv_comm_unique_id produced on one rank, carried to
 * the others out of band (torch.distributed, MPI, a file).  world == 1 with id128 == NULL needs no RCCL. */
typedef struct kv_comm* kv_comm_t;
int kv_comm_unique_id(void* id128);
int kv_comm_create(int world, int rank, const void* id128, int device, kv_comm_t* out);
/* A communicator whose exchanges the CALLER makes on the host (a rehearsal of the N > 1 control flow where RCCL cannot
 * run: several ranks sharing one GPU, a CPU-side transport in a test).  Same ops, same agreement and failure protocol
 * as over RCCL; the library synchronises its stream, then calls
 *   exchange(user, send_dev, recv_dev, bytes_per_peer): segment p of send_dev to rank p, segment q of recv_dev from
 *     rank q (device pointers, [world][bytes_per_peer]); returns 0 when the data has arrived;
 *   max_u32(user, value): *value = the maximum of *value over all ranks (the lossless agreement); may be NULL when no
 *     shard used with the communicator is lossless.
 * Every rank must call the same ops in the same order, as with RCCL.  Never a measurement path. */
typedef int (*kv_comm_exchange_fn)(void* user, const void* send_dev, void* recv_dev, int64_t bytes_per_peer);
typedef int (*kv_comm_max_fn)(void* user, uint32_t* value);
int kv_comm_create_staged(int world, int rank, kv_comm_exchange_fn exchange, kv_comm_max_fn max_u32, void* user,
                          int device, kv_comm_t* out);
int kv_comm_destroy(kv_comm_t comm);
/* Ops of one table run in issue order whatever their streams: when the stream changes, the table's next op first waits
 * for the previous one (an event recorded on the PREVIOUS stream).  A stream handed to an op must therefore still exist at
 * the table's next op.  kv_comm_destroy retires the communicator's own stream from every table by itself; a caller that
 * destroys a stream of its own synchronises it and says so here (the call synchronises `stream`, then no table refers to
 * it any more). */
int kv_forget_stream(kv_stream_t stream);
/* the communicator's stream: a caller that passes it as `stream` to kv_shard_lookup / kv_shard_apply (its other work
 * queued there too) pays no event hop into and out of the exchange */
int kv_comm_stream(kv_comm_t comm, kv_stream_t* stream);
/* bytes_per_peer bytes to and from every rank (send / recv: [world][bytes_per_peer] device buffers), ordered after
 * what `stream` has been given and before what it is given next (events; no host synchronisation). */
int kv_comm_all_to_all(kv_comm_t comm, const void* send, void* recv, int64_t bytes_per_peer, kv_stream_t stream);

/* This rank's side of a table sharded over `world` ranks: `local_table` holds the rows of the ids this rank owns
 * (optimizer slot tables are sharded the same way: same rule, same rank).  max_ids = largest batch (<= 2^21);
 * peer_capacity = (id, count) records per peer in the fixed-capacity exchange buffers (0 = twice an even share of
 * max_ids); it must be the same on every rank, like world, dim and owner_rule (checked by the first exchange:
 * KV_FAILED_PRECONDITION).  A batch that needs more for one owner: its surplus ids read zeros and their gradients
 * are dropped; the next kv_shard_lookup / kv_shard_lookup_route returns KV_RESOURCE_EXHAUSTED AFTER queuing all its
 * own work (a rank that reports stays in step with its peers).  Raise peer_capacity on every rank.
 *
 * One training step per rank, all on the device, no host synchronisation, no size exchange:
 *   lookup  kv_shard_lookup_route   ids -> distinct ids + occurrence counts -> the owners' segments of send_pairs
 *           [exchange of the records]
 *           kv_shard_lookup_serve   the owner looks the received ids up in its table (GatherOrInsertWithCounts:
 *                                   frequency words count every occurrence) -> send_rows, record for record
 *           [exchange of the rows]
 *           kv_shard_lookup_finish  out[i] = the row that came back for ids[i]
 *   apply   kv_shard_apply_route    gradients summed per distinct id into the records their ids were sent in
 *           [exchange of the rows]
 *           kv_shard_apply_serve    the owner's fused optimizer apply, on the index its lookup left behind
 * kv_shard_lookup / kv_shard_apply chain the phases with the RCCL exchanges on the communicator's stream, forked from
 * `stream` by an event: the caller's stream stays free for the dense tower until kv_shard_join (or join != 0).
 * With `stream` == kv_comm_stream(comm) nothing is forked or joined (one queue, no event hops).  This rank's own
 * segment of every exchange stays where it is — the phases the whole ops run read it in the send buffers (a caller
 * that chains the phase calls itself moves all segments, its own included, into the receive buffers: a device copy) —
 * the others are one grouped ncclSend / ncclRecv pair per peer. */
typedef struct kv_shard* kv_shard_t;
int kv_shard_create(kv_handle_t local_table, int world, int rank, int owner_rule, int64_t max_ids,
                    int64_t peer_capacity, kv_shard_t* out);
int kv_shard_destroy(kv_shard_t shard);
int kv_shard_buffers(kv_shard_t shard, void** send_pairs, void** recv_pairs, void** send_rows, void** recv_rows,
                     int64_t* pair_bytes_per_peer, int64_t* row_bytes_per_peer);
int kv_shard_lookup_route(kv_shard_t shard, const void* ids, int64_t n, kv_stream_t stream);
int kv_shard_lookup_serve(kv_shard_t shard, kv_stream_t stream);
int kv_shard_lookup_finish(kv_shard_t shard, float* out, kv_stream_t stream);
int kv_shard_apply_route(kv_shard_t shard, const float* grad, kv_stream_t stream);
/* optimizer: 0 GroupAdam V4, 1 GroupAdam V3 (hp = lr, beta1_power, beta2_power, beta1, beta2, epsilon, l1, l2, l21),
 * 2 Adagrad (hp = lr, update_slots), 3 SparseGroupFtrl (hp = lr, l1, l2, l21, l2_shrinkage, lr_power; slot1 = linear) */
int kv_shard_apply_serve(kv_shard_t shard, int optimizer, kv_handle_t slot0, kv_handle_t slot1, const float* hp,
                         kv_stream_t stream);
int kv_shard_lookup(kv_shard_t shard, kv_comm_t comm, const void* ids, int64_t n, float* out, int join,
                    kv_stream_t stream);
int kv_shard_apply(kv_shard_t shard, kv_comm_t comm, int optimizer, kv_handle_t slot0, kv_handle_t slot1,
                   const float* grad, const float* hp, int join, kv_stream_t stream);
int kv_shard_join(kv_shard_t shard, kv_stream_t stream);
/* Lossless mode — ON by default since round 4: a sharded lookup can lose nothing, like the reference's capacity-free
 * partitioned lookup (embedding_ops.py:115-204).  kv_shard_lookup / kv_multi_shard_lookup first agree on the largest
 * segment any rank's route produced (one 4-byte ncclAllReduce per table and ONE stream synchronisation per call); when
 * it exceeds the capacity every rank raises its capacity to the same new value and routes again, so nothing is ever
 * dropped and the ranks stay in step.  kv_shard_set_lossless(shard, 0) — on every rank alike — opts into the
 * synchronisation-free mode for deployments that size peer_capacity themselves: no host round trip (about 15 us per
 * step), but a batch that overflows peer_capacity loses its surplus ids (zero rows, dropped gradients) and says so
 * one call late (KV_RESOURCE_EXHAUSTED).  The phase calls (kv_shard_lookup_route / _serve / _finish ...) are building
 * blocks for a caller with its own exchange and never synchronise: such a caller checks kv_shard_agree_local or sizes
 * the capacity. */
int kv_shard_set_lossless(kv_shard_t shard, int on);

/* Where a sharded step's time goes (for the first multi-GPU run, where nothing can be iterated on): every `every`-th
 * kv_shard_lookup / kv_shard_apply records events on the communicator's stream at its phase boundaries (the calls that
 * do not are unchanged; every = 0 switches it off and clears the sums).  kv_shard_profile_read gives, per phase, the
 * milliseconds summed over the samples and their number — phases in step order (embedding_ops.py:115-204: partition ->
 * gather -> stitch, and the gradient's way back):
 *   0 route          ids -> local distinct ids -> the owners' segments (+ the lossless agreement when it is on)
 *   1 exchange_ids   grouped ncclSend / ncclRecv of the (id, count) segments
 *   2 serve          the owner's training lookup over the segments it received
 *   3 exchange_rows  the rows back
 *   4 finish         out[i] = the row that came back for ids[i]
 *   5 presum         gradient rows summed per distinct id into the records the ids were sent in
 *   6 exchange_grads the summed rows to the owners
 *   7 apply          the owner's fused optimizer apply
 * info (may be NULL) [4]: {ranks whose handshake record arrived in the first exchange, peer capacity now, times the
 * capacity was raised (lossless mode), batches that overflowed a segment (lossy mode)}. */
#define KV_SHARD_PHASES 8
int kv_shard_profile(kv_shard_t shard, int every);
int kv_shard_profile_read(kv_shard_t shard, double* ms_sum, int64_t* samples, int n_phases, int64_t* info);
/* The same agreement between the shards of ONE process (kv_shard_exchange_local's companion): after every shard's
 * kv_shard_lookup_route; *rerouted != 0: the capacity was raised on all of them, run the routes again. */
int kv_shard_agree_local(const kv_shard_t* shards, int world, int* rerouted, kv_stream_t stream);
/* Several sharded tables in one step (the embedding tables of one model; the reference looks its tables up one op
 * each, python/ops/embedding_ops.py:150-204 per variable): all route phases, ONE grouped exchange carrying every
 * table's segments, all serve phases, ONE exchange back, all finish phases — two exchanges per lookup and one per
 * apply whatever ntab is.  Results equal the per-table ops'.  ids[k] / n[k] / outs[k] / grads[k] / slot0[k] belong to
 * shards[k]; slot1 may be NULL unless optimizer == 3.  A deferred join (join == 0) is completed by
 * kv_shard_join(shards[0], stream). */
int kv_multi_shard_lookup(const kv_shard_t* shards, int ntab, kv_comm_t comm, const void* const* ids, const int64_t* n,
                          float* const* outs, int join, kv_stream_t stream);
int kv_multi_shard_apply(const kv_shard_t* shards, int ntab, kv_comm_t comm, int optimizer, const kv_handle_t* slot0,
                         const kv_handle_t* slot1, const float* const* grads, const float* hp, int join,
                         kv_stream_t stream);
/* the exchange between shards of ONE process on one device (segment r of shard p's send buffer -> segment p of
 * shard r's receive buffer; what: 0 records, 1 rows): several ranks on a single GPU, where RCCL cannot be used */
int kv_shard_exchange_local(const kv_shard_t* shards, int world, int what, kv_stream_t stream);

/* Row permutation for the exchange (no handle: plain device buffers).  scatter == 0:
 * out[i] = src[index[i]]; scatter == 1: out[index[i]] = src[i]; rows of row_bytes (a multiple of
 * 4) bytes, index [n] int32; index_outer (may be NULL, gather only): out[i] =
 * src[index[index_outer[i]]] for i < n = len(index_outer).  Replaces the tf.gather / tf.dynamic_stitch the reference's
 * partitioned lookup uses to undo its `ids % num_shards` split (python/ops/embedding_ops.py:
 * 150-204). */
int kv_take_rows(int device, const void* src, const int32_t* index, const int32_t* index_outer, int64_t n,
                 int64_t row_bytes, int scatter, void* out, kv_stream_t stream);

/* ---- measurement hooks (no reference counterpart; the reference only VLOGs wall time,
 * kernels/training_ops.cc:6989,7211) -----------------------------------------------------------
 * kv_profile_enable(h, max_launches > 0) brackets every kernel this table launches with a pair
 * of hipEvents on the op's own stream (0 turns it off and frees the events).  kv_profile_read
 * (synchronous) sums the elapsed milliseconds per kernel kind since the last read.
 * For the optimizer ops the events belong to the `var` table.  kv_profile_select(h, mask) limits
 * the bracketing to the kinds whose bit (1 << KV_PROF_*) is set (default: all) — an event pair
 * costs a few microseconds of stream time, so a throughput measurement brackets one kernel. */
#define KV_PROF_LOOKUP_TILE 0   /* k_tile / k_ltile: tile dedup + partition sort (k_ltile: + index probes + output rows) */
#define KV_PROF_LOOKUP_PART 1   /* k_part_keys<LOOKUP> / k_part2: find / insert / frequency, key records, work items */
#define KV_PROF_LOOKUP_ORDER 2  /* k_gather<ORDER>: output rows + the sorted position list */
#define KV_PROF_INDEX 3         /* the three index kernels of an apply that was not handed a valid token */
#define KV_PROF_APPLY_SORTED 4  /* k_apply: segmented gradient sum + fused row update */
#define KV_PROF_APPLY_SPAN 5    /* k_apply_fin: keys that span several chunks */
#define KV_PROF_APPLY_TSUM 6    /* k_tsum: tile-local gradient sums of repeated ids (entry-list pipeline) */
#define KV_PROF_APPLY_UNIQUE 7  /* k_uapply: the optimizer apply on unique ids + pre-summed rows (kv_apply_*_unique), one launch */
#define KV_PROF_APPLY_TILE 8    /* k_ltsum: the batch's tile pass + the tile sums in front of the optimizer apply */
#define KV_PROF_KINDS 9
int kv_profile_enable(kv_handle_t h, int max_launches);
int kv_profile_read(kv_handle_t h, double* ms_sum, int64_t* launches, int n_kinds);
int kv_profile_select(kv_handle_t h, unsigned kind_mask);
int kv_profile_sample(kv_handle_t h, int every);

#ifdef __cplusplus
}
#endif
#endif /* KVHIP_H_ */
