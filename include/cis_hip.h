/*
 * cis_hip.h -- C ABI of libcis_hip.so: the MI355X (gfx950) implementation of the
 * embed-then-index hot path of ColumbiaImageSearch.
 *
 * The reference has no FFI: its boundary is three duck-typed Python surfaces
 * (lopq.model.LOPQModel[PCA], lopq.search.LOPQSearcherBase subclasses, cufacesearch
 * GenericFeaturizer).  The Python mirrors of those surfaces in columbiaimagesearch_amd/ bind
 * exactly the entry points below through ctypes; INTEGRATION.md shows the binding a reference
 * maintainer would add.  Each entry point cites the reference code it replaces
 * (paths relative to the reference root).
 *
 * Conventions
 *  - plain pointers and sizes only; `*_dev` entry points take DEVICE pointers and a hipStream_t
 *    (passed as void*; NULL = the null stream) and never synchronise; the others take HOST
 *    pointers, copy, run and synchronise before returning;
 *  - every function returns 0 on success or a negative CIS_E* code; the message of the last
 *    failure on the calling thread is cis_last_error().  Nothing aborts or throws across the ABI
 *    (the reference's callers log per-item errors and keep going: lopq/lopq/search.py:365-367);
 *  - the library never keeps caller pointers after a call returns; handles own device memory;
 *  - HIP is initialised lazily on first use, so a process may fork before touching the library
 *    (gunicorn / multiprocessing callers); handles must not be shared across processes;
 *  - a handle (model, index, CNN) owns ONE set of device workspaces: calls on the same handle must not overlap -- neither from
 *    two threads nor on two streams (a `*_dev` call returns while its kernels are still queued: issue the next call on the
 *    same stream, or wait for the first).  An index also uses its model's workspaces.  Different handles are independent;
 *  - dtype arguments: CIS_F32 = 4, CIS_F64 = 8 (bytes per element).
 */
#ifndef CIS_HIP_H
#define CIS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CIS_F32 4
#define CIS_F64 8

#define CIS_OK 0
#define CIS_EINVAL -1   /* bad argument (Python wrapper raises ValueError)            */
#define CIS_EHIP -2     /* a HIP runtime call failed                                   */
#define CIS_ENOMEM -3   /* device or host allocation failed                            */
#define CIS_EUNSUPPORTED -4 /* valid in the reference but not built yet (NotImplementedError) */
#define CIS_ENODEVICE -5 /* no gfx950 device visible                                   */

typedef struct cis_model cis_model;
typedef struct cis_index cis_index;
typedef struct cis_cnn cis_cnn;

/* One ranked candidate of a (possibly partial, per-shard) result list.  Ordering key is
 * (dist, visit_rank, pos): the reference ranks with a stable sort over the retrieval order,
 * i.e. multisequence cell order then insertion order inside the cell (lopq/lopq/search.py:128-133,
 * :210).  32 bytes; this is also what travels in the RCCL all-gather of the sharded search. */
typedef struct cis_hit {
    double dist;         /* squared ADC distance, float64 (search.py:173)                 */
    uint32_t visit_rank; /* 0-based position of the candidate's cell in multisequence order */
    uint32_t pos;        /* 0-based insertion position inside that cell                   */
    int64_t id;          /* caller's item id (-1: empty slot)                             */
    int32_t cell;        /* c0 * V + c1 of the candidate's coarse cell                    */
    int32_t reserved;
} cis_hit;

/* ---- library ------------------------------------------------------------------------------ */
int cis_version(void);
const char* cis_last_error(void);
/* Device workspace growth since the process started: every handle's workspaces are grow-only, and a growth is a hipFree (which
 * waits for the device) + hipMalloc -- tens of milliseconds when it lands inside a measured loop.  A harness reads the counters
 * before and after its timed region to prove that nothing was allocated in it (bench.py does).  Either pointer may be NULL. */
int cis_alloc_stats(int64_t* n_allocs, int64_t* n_bytes);
/* Number of visible HIP devices (0 when none); never fails. */
int cis_device_count(void);
/* Select the device used by handles created afterwards by this process (default 0). */
int cis_set_device(int device);
/* Runs a device self test of the wave-level primitives the scan kernel relies on (DPP / permlane
 * lane exchanges, in-register bitonic sort); *n_errors = 0 when they behave. */
int cis_selftest(int* n_errors);

/* ---- LOPQ model: replaces lopq.model.LOPQModel / LOPQModelPCA arithmetic --------------------
 * Parameter layout = the reference's parameter tuple (lopq/lopq/model.py:461-473, :841), each
 * pair concatenated split-major and C-contiguous:
 *   Cs   [2][V][h]      coarse centroids, dtype coarse_dtype (float32 when LOPQ was trained on
 *                       float32 data, e.g. after apply_PCA; float64 otherwise)
 *   Rs   [2][V][h][h]   local rotations, applied as R[c] . r   (model.py:638)
 *   mus  [2][V][h]      mean residuals
 *   subs [M][K][w]      sub-quantizer centroids, fine split j of coarse split s at s*M/2 + j
 *   pca_P [D_in][D], pca_mu [D_in]  or NULL/NULL for a plain LOPQModel (then D_in == D)
 * with h = D/2, w = D/M.  K <= 256 and V <= 4096 in this build.
 * pca_mu_dtype: dtype the caller's pca_mu array had.  train_pca takes np.mean of the training
 * data (model.py:260), so a model trained on float32 features holds a float32 mean and
 * apply_PCA's `x - pca_mu` (model.py:965) then rounds in float32 for float32 inputs; the values
 * are passed here widened to double either way. */
int cis_model_create(cis_model** out, int D_in, int D, int V, int M, int K, int coarse_dtype,
                     const void* Cs, const double* Rs, const double* mus, const double* subs,
                     const double* pca_P, const double* pca_mu, int pca_mu_dtype, int renorm);
void cis_model_destroy(cis_model* m);

/* LOPQModelPCA.apply_PCA (model.py:961-978): out[n][D] float32.  x_dtype = dtype of X. */
int cis_apply_pca(cis_model* m, const void* X, int x_dtype, int64_t n, float* out);

/* LOPQModel.predict over n vectors (model.py:543-561, :980-1003; the loop of
 * compute_codes_notparallel, lopq/lopq/utils.py:203-218).  X is [n][D_in] of x_dtype (PCA is
 * applied first when the model has one).  coarse [n][2], fine [n][M]. */
int cis_encode(cis_model* m, const void* X, int x_dtype, int64_t n, uint16_t* coarse, uint8_t* fine);
int cis_encode_dev(cis_model* m, const void* dX, int x_dtype, int64_t n, uint16_t* d_coarse,
                   uint8_t* d_fine, void* stream);

/* The pieces of predict, exposed because the reference exposes them.  They take vectors that
 * are ALREADY in LOPQ space (i.e. after apply_PCA), [n][D] of x_dtype:
 *   predict_coarse (model.py:563-573), project (model.py:604-641) -> out [n][D] float64,
 *   predict_fine (model.py:575-602),
 *   get_subquantizer_distances (model.py:673-704) -> tables [n][M][K] float64,
 *   reconstruct (model.py:643-671) -> out [n][D] float64. */
int cis_predict_coarse(cis_model* m, const void* X, int x_dtype, int64_t n, uint16_t* coarse);
int cis_project(cis_model* m, const void* X, int x_dtype, int64_t n, const uint16_t* coarse, double* out);
int cis_predict_fine(cis_model* m, const void* X, int x_dtype, int64_t n, const uint16_t* coarse, uint8_t* fine);
int cis_subquantizer_distances(cis_model* m, const void* X, int x_dtype, int64_t n,
                               const uint16_t* coarse, double* tables);
int cis_reconstruct(cis_model* m, const uint16_t* coarse, const uint8_t* fine, int64_t n, double* out);

/* predict_cluster (lopq/lopq/utils.py:33-53) against an arbitrary centroid matrix: X [n][d], C [ncent][d];
 * distances in float32 when both are float32, else float64 (numpy promotion), numpy summation order, first
 * minimum wins.  out [n] cluster ids. */
int cis_predict_cluster(const void* X, int x_dtype, const void* C, int c_dtype, int64_t n, int ncent, int d,
                        uint32_t* out);

/* multisequence (lopq/lopq/search.py:13-82) as a list: for each of n LOPQ-space vectors X [n][2h] the first
 * max_cells (clamped to V*V) cells in multi-sequence order: cells [n][max_cells][2], dists [n][max_cells]
 * (the cell distance d0+d1 rounded in *dist_dtype = 4 or 8, the dtype the reference yields). */
int cis_multisequence(const void* X, int x_dtype, const void* C0, const void* C1, int c_dtype, int64_t n, int V,
                      int h, int max_cells, int32_t* cells, double* dists, int* dist_dtype);

/* ---- LOPQ index: replaces lopq.search.LOPQSearcher (search.py:310-382) ---------------------- */
/* The index keeps a borrowed pointer to `m`: destroy the index first. */
int cis_index_create(cis_index** out, cis_model* m);
void cis_index_destroy(cis_index* ix);
/* A search VIEW of `base`: shares its storage (codes, ids, offsets, cell sizes -- lopq/lopq/search.py:310-382's `index` dict) and
 * owns only per-batch workspaces and counters, so that two query batches can be in flight at once, each handle on its own stream
 * (the reference answers independent queries from 16 independent gunicorn workers over one LMDB index, searcher_lopqhbase.py:198-206:
 * this is the same sharing inside one process).  A view is read-only (inserts / cell reads go to the base) and should be destroyed
 * before the base: a base destroyed first ORPHANS its views -- they stay valid handles, every later search through them returns
 * CIS_EINVAL, and they must still be destroyed.  Inserts into the base must be stream-ordered against the views' searches by the
 * caller (columbiaimagesearch_amd/distributed.py:ShardedSearcher._insert_fence does it for its own lanes). */
int cis_index_create_view(cis_index** out, cis_index* base);

/* Cell-sharded operation (one process per GPU): this handle stores only the cells with
 * owner[cell] == rank but counts every cell, so that all ranks derive the same global
 * multisequence order and quota cut-off without communicating (search.py:128-133).
 * owner = NULL selects cell_id % world.  Must be called on an empty index. */
int cis_index_set_shard(cis_index* ix, int rank, int world, const int32_t* owner /* [V*V] or NULL */);
/* How the insert calls of this handle were served: counters[0] = batches written IN PLACE behind their cells' last items (O(batch):
 * the reference appends to a per-cell list, lopq/lopq/search.py:349-364), counters[1] = batches that rebuilt the layout (a cell's
 * slack was exhausted, or a bulk load). */
int cis_index_insert_counters(cis_index* ix, int64_t counters[2]);

/* Routed insert into a cell-sharded index: cis_index_add is then given only the codes of
 * the cells this rank owns; cis_index_cell_counts reads the per-cell sizes [V*V] (all shards), and
 * cis_index_add_remote_counts adds per-cell increments [V*V] for the cells owned by OTHER ranks (entries of owned cells
 * are ignored), so that every rank keeps the whole cell-size table the quota cut needs (lopq/lopq/search.py:128-133). */
int cis_index_cell_counts(cis_index* ix, int64_t* counts);
int cis_index_add_remote_counts(cis_index* ix, const int64_t* delta);
/* The same on device arrays (the all-reduce of the increments runs over RCCL on them): d_counts / d_delta [V*V]. */
int cis_index_cell_counts_dev(cis_index* ix, int64_t* d_counts, void* stream);
int cis_index_add_remote_counts_dev(cis_index* ix, const int64_t* d_delta, void* stream);
/* Routed insert without a host copy.  cis_index_route_pack_dev groups this rank's n freshly encoded items by the rank
 * that owns their cell (arrival order inside a group, so that the per-cell insertion order after the exchange is that of
 * a single index, search.py:359): d_records [n][12 + M] = id (8 B, little endian) | coarse (2 x uint16) | fine (M B),
 * d_counts [world] int64 = records per destination.  cis_index_add_records_dev inserts records received from the
 * all-to-all (same semantics and outputs as cis_index_add_dev). */
int cis_index_route_pack_dev(cis_index* ix, const int64_t* d_ids, const uint16_t* d_coarse, const uint8_t* d_fine,
                             int64_t n, uint8_t* d_records, int64_t* d_counts, void* stream);
int cis_index_add_records_dev(cis_index* ix, const uint8_t* d_records, int64_t n, int dedup, int64_t* n_added,
                              int64_t* n_invalid, int64_t* d_cell_delta, void* stream);

/* add_codes (search.py:325-369): append items in order; with dedup != 0 an id already present in
 * the SAME cell is skipped (first occurrence wins).  *n_added = items counted (all shards). */
int cis_index_add(cis_index* ix, const int64_t* ids, const uint16_t* coarse, const uint8_t* fine,
                  int64_t n, int dedup, int64_t* n_added);
/* The same insert on arrays that are already in HBM (the codes cis_encode_dev just produced: the refresh loop of
 * searcher_lopqhbase.py:743-758 without the round trip through the host).  The index lives in HBM only; an insert is a
 * stable merge by kernels on `stream` (csrc/lopq_index.hip), and the call returns after the stream has drained (the
 * accepted count is read back).  Items with out-of-range codes or negative ids are skipped and counted in *n_invalid
 * ("could not push code", search.py:365-367) -- the host entry point above rejects the whole call instead. */
int cis_index_add_dev(cis_index* ix, const int64_t* d_ids, const uint16_t* d_coarse, const uint8_t* d_fine, int64_t n,
                      int dedup, int64_t* n_added, int64_t* n_invalid, int64_t* d_cell_delta /* [V*V] accepted per cell, or NULL */,
                      void* stream);
/* featsio.normfeatB64encode's normalisation (cufacesearch/cufacesearch/featurizer/featsio.py:13-22) on n device rows of d
 * values, in place, in the rows' dtype; zero rows stay zero. */
int cis_l2_normalize_dev(void* d_x, int dtype, int64_t n, int d, void* stream);
/* get_nb_indexed (search.py:91-92): items over all shards. */
int64_t cis_index_size(cis_index* ix);
/* get_cell (search.py:372-382): items of one cell in insertion order.  cap <= 0: *n = the cell's size (all
 * shards), nothing is copied.  cap > 0: copies at most cap items, *n = the number copied (0 for a cell that
 * another shard owns). */
int cis_index_get_cell(cis_index* ix, int c0, int c1, int64_t cap, int64_t* ids, uint8_t* fine, int64_t* n);

/* LOPQSearcherBase.search over nq queries (search.py:179-224): Q [nq][D_in] of q_dtype.
 * limit < 0 means "limit = quota" (search.py:213-214).  Outputs, with L = effective limit:
 *   ids [nq][L] (-1 padded), dists [nq][L] (NaN padded), n_found [nq], visited [nq], and
 *   optionally (may be NULL) cells [nq][L] (c0*V+c1, -1 padded) and pos [nq][L]: where each
 *   result lives, so that its code can be fetched with cis_index_get_codes (Result.code,
 *   search.py:217-222). */
int cis_index_search(cis_index* ix, const void* Q, int q_dtype, int nq, int64_t quota, int limit,
                     int64_t* ids, double* dists, int32_t* n_found, int32_t* visited,
                     int32_t* cells, uint32_t* pos);
int cis_index_search_dev(cis_index* ix, const void* dQ, int q_dtype, int nq, int64_t quota, int limit,
                         int64_t* d_ids, double* d_dists, int32_t* d_n_found, int32_t* d_visited,
                         int32_t* d_cells, uint32_t* d_pos, void* stream);
/* The host-pointer search in two halves (round 5): cis_index_search_async enqueues copy-in, search and copy-out on the handle's OWN
 * stream and returns once the search's launches are queued; cis_index_search_wait blocks until the results have landed in the caller's
 * buffers, which must stay valid and untouched until then.  One batch in flight per handle (a second call waits for the first); views
 * of one index (cis_index_create_view) give several -- the reference serves independent queries from 16 gunicorn workers over one
 * index (searcher_lopqhbase.py:198-206), this is that inside one process.  cis_index_search = async + wait.
 * cis_host_alloc / cis_host_free: page-locked host memory.  With Q and the outputs in it the copies are DMA transfers that overlap the
 * other handles' searches; pageable memory works too and is staged by the runtime (slower, and the call then blocks while it copies). */
int cis_index_search_async(cis_index* ix, const void* Q, int q_dtype, int nq, int64_t quota, int limit,
                           int64_t* ids, double* dists, int32_t* n_found, int32_t* visited,
                           int32_t* cells, uint32_t* pos);
int cis_index_search_wait(cis_index* ix);
int cis_host_alloc(void** out, size_t bytes);
void cis_host_free(void* p);
/* Fine codes of n stored items addressed by (cell, pos) as returned by a search.  fine [n][M].
 * Items of cells owned by another shard yield CIS_EINVAL. */
int cis_index_get_codes(cis_index* ix, const int32_t* cells, const uint32_t* pos, int64_t n, uint8_t* fine);

/* Sharded search, step 1: this shard's ranked candidates.  d_hits [nq][L] (unused slots have
 * id = -1, dist = +inf), d_visited [nq] (identical on every rank). */
int cis_index_search_partial_dev(cis_index* ix, const void* dQ, int q_dtype, int nq, int64_t quota,
                                 int limit, cis_hit* d_hits, int32_t* d_visited, void* stream);
/* The same, packed for the exchange: d_cnt[q] valid hits of query q at d_packed[d_off[q] ..], queries in order,
 * *d_total hits in all (d_packed needs room for nq * L records; only the first *d_total are written). */
int cis_index_search_partial_packed_dev(cis_index* ix, const void* dQ, int q_dtype, int nq, int64_t quota, int limit,
                                        cis_hit* d_packed, int32_t* d_cnt, int64_t* d_off, int64_t* d_total,
                                        int32_t* d_visited, void* stream);
/* Routed cell-sharded search (round 5; the all-gather protocol above replicates projection, cell ranking and walk on every rank):
 * step 0, on a query's HOME rank -- d_mask[q] bit r = rank r owns a non-empty cell among those the multisequence walk of query q
 * visits before the quota is reached (the walk of lopq/lopq/search.py:58-82,:128-133 against the cell sizes of the WHOLE index;
 * world <= 64), d_visited[q] (or NULL) = the number of cells it looks at (the `visited` of the reference's result).  Uses the
 * handle's workspaces: call it on the stream the handle's searches run on. */
int cis_index_query_owners_dev(cis_index* ix, const void* d_q, int q_dtype, int nq, int64_t quota, uint64_t* d_mask,
                               int32_t* d_visited, void* stream);
/* ... step 1: the send buffers of the query all-to-all.  d_q [nq] rows of row_bytes bytes (a multiple of 4: float32 or float64
 * queries), d_slot [world][nq]: row of query i in the block for rank d (-1: not sent; rows in query order), d_out_q [world][cap]
 * rows, d_cnt [world]: rows used per destination, *d_overflow = 1 when a destination needed more than cap rows (the caller then
 * answers the batch through the all-gather protocol). */
int cis_route_queries_dev(const void* d_q, int nq, int row_bytes, const uint64_t* d_mask, int world, int cap, void* d_out_q,
                          int32_t* d_slot, int32_t* d_cnt, int32_t* d_overflow, void* stream);

/* ... step 3, back on the home rank: where the list of home query i from rank d sits in the buffer of returned lists (rows of
 * `limit` cis_hit, grouped by answering rank; h_base [world] (HOST) = first row of rank d's group): d_off [world][nq] record
 * offsets, d_cnt [world][nq] valid hits (0: rank d was not asked) -- the inputs of cis_merge_packed_dev with stride = 0. */
int cis_routed_merge_tables_dev(const int32_t* d_slot, int world, int nq, const int64_t* h_base, const cis_hit* d_hits, int limit,
                                int64_t* d_off, int32_t* d_cnt, void* stream);

/* Sharded search, step 2 (after the all-gather): merge `world` partial lists
 * d_parts [world][nq][L] into the final ranking. */
int cis_merge_hits_dev(const cis_hit* d_parts, int world, int nq, int limit, int64_t* d_ids,
                       double* d_dists, int32_t* d_n_found, int32_t* d_cells /* or NULL */,
                       uint32_t* d_pos /* or NULL */, void* stream);

/* The same for PACKED partial lists (what travels over xGMI): shard w contributed its valid hits only, in query
 * order, d_parts[w*stride + d_off[w*nq + q] .. + d_cnt[w*nq + q]) for query q (stride = 0: ONE buffer, d_off holds absolute record
 * offsets -- the return trip of the routed search).  limit <= 3072 (one wave per query; above that the caller ranks the packed lists itself: distributed.merge_packed_sorted). */
int cis_merge_packed_dev(const cis_hit* d_parts, int world, int64_t stride, const int64_t* d_off /* [world][nq] */,
                         const int32_t* d_cnt /* [world][nq] */, int nq, int limit, int64_t* d_ids, double* d_dists,
                         int32_t* d_n_found, int32_t* d_cells /* or NULL */, uint32_t* d_pos /* or NULL */, void* stream);

/* Offsets of the packed exchange, on the device ("one collective per query batch" -- no host read in between):
 * d_cnt_all [world][nq] = the all-gathered per-query hit counts -> d_off [world][nq] (exclusive scan per shard), d_totals [world],
 * *d_overflow = 1 when a shard holds more records than `stride` (the fixed per-rank size of the payload all-gather). */
int cis_exchange_offsets_dev(const int32_t* d_cnt_all, int world, int nq, int64_t stride, int64_t* d_off, int64_t* d_totals,
                             int32_t* d_overflow, void* stream);

/* Exact re-ranking with features resident in HBM (cufacesearch/cufacesearch/searcher/searcher_lopqhbase.py:864-912:
 * dist = np.linalg.norm(normed_feat - res_fts[pos]) for the first results of a query).  d_feats [n_feats][D] and
 * d_q [nq][D] in f_dtype (4 = float32, 8 = float64; the distance is computed in that type, NOT squared), d_rows [nq][L]
 * = feature row of each result (-1: not resident -> NaN, the caller keeps the ADC distance as the reference does). */
int cis_rerank_dev(const void* d_feats, int f_dtype, int64_t n_feats, int D, const void* d_q, int nq,
                   const int64_t* d_rows, int L, double* d_dists, void* stream);

/* Lloyd iterations for training the coarse and fine codebooks (lopq/lopq/model.py:339-437 uses scikit-learn k-means;
 * k-means is not bit-reproducible across libraries, so this is judged by distortion).  X [n][d] float32 (host),
 * centroids [k][d] float32: initial centroids in, trained centroids out; k * d <= 7680.  assign [n] (or NULL) and *inertia
 * (sum of squared distances, or NULL) refer to the returned centroids. */
int cis_kmeans(const float* X, int64_t n, int d, int k, int iters, float* centroids, int32_t* assign, double* inertia);

/* Training accumulations on the GPU (csrc/lopq_train.hip), float64, host pointers.  Rows of X [n][d] are sorted by group;
 * group_off [groups+1] are the row offsets (group_off[0] = 0, group_off[groups] = n).
 * cis_train_gram:    G[g] = sum_{r in g} x_r x_r^T [groups][d][d] and S[g] = sum x_r [groups][d] (S may be NULL): the np.outer
 *                    accumulators of the PCA covariance (lopq/lopq/model.py:263-267, one group) and of the per-cluster
 *                    residual covariances (:142-155).
 * cis_train_project: Y[r] = R[g] . (x_r - mu[g]) for the rows of group g (compute_local_rotations' projection, :209-234);
 *                    R [groups][d][d] row-major, mu [groups][d]. */
int cis_train_gram(const double* X, int64_t n, int d, const int64_t* group_off, int groups, double* G, double* S);
int cis_train_project(const double* X, int64_t n, int d, const int64_t* group_off, int groups, const double* R,
                      const double* mu, double* Y);

/* Counters of the last search on this handle (for bench.py's roofline):
 *   stats[0] candidates scanned (sum over queries of retrieved items on this shard)
 *   stats[1] (query, cell) work items   stats[2] ADC tables built   stats[3] scan kernel launches */
int cis_index_last_stats(cis_index* ix, int64_t stats[4]);
/* Which scan kernel the last batch of this handle ran (bench.py names it in `roofline.kernel`): 0 none (all-candidates path),
 * 1 k_adc_scan (float64), 2 k_adc_scan2 (float32 prefilter), 3 k_adc_scan3 (16-bit fixed-point prefilter: streaming / two-pass form),
 * 4 k_adc_scan4 (the sampled single-pass form of the 16-bit prefilter; k_adc_scan3 then only works off its fall-back slots). */
int cis_index_last_scan_kernel(cis_index* ix);

/* Stage timing with HIP events recorded on the stream the kernels are launched on (bench.py's
 * roofline).  While enabled every search records events around its stages; read_profile waits for
 * them, returns the accumulated milliseconds since the previous read and clears the accumulators:
 *   ms[0] front end (PCA, coarse distances, rank, multisequence plan)   ms[1] ADC tables
 *   ms[2] ADC scan stage (slot list + scan kernel)                        ms[3] per-query merge
 *   ms[4] the ADC scan kernel alone (events right before and after its launch)
 *   *launches = number of scan kernel launches accumulated. */
int cis_index_set_profiling(cis_index* ix, int level /* 0 off, 1 only the pair of events around the scan kernel (ms[4]), 2 every stage */);
/* Ranking route selection: 0 = automatic (limit <= 952: a prefilter scan kernel with exact float64 re-scoring of its
 * survivors -- the 16-bit fixed-point kernel k_adc_scan3 for batches of >= 256 queries over short cells (< 8192 codes on
 * average, M <= 8, limit <= 440), the float32 kernel k_adc_scan2 otherwise; small batches, limit > 952 and indexes of tiny cells
 * (thousands of coarse clusters) take the all-candidates path instead: exact distances of every candidate, radix select;
 * exact float64 scan kernel otherwise), 1 = exact float64 scan kernel wherever it applies (limit <= 3072),
 * 2 = the float32-prefilter kernel for every batch size, 3 / 4 = the 16-bit fixed-point kernel for every batch size in its
 * streaming / two-pass (histogram threshold, then collection) form, 5 = the same kernel family's sampled single-pass form
 * (threshold from a sample of the chunk, verified after the pass; what large batches over short cells take by default),
 * 6 = the HBM-streaming route (csrc/lopq_stream.hip: what batches of <= 16 queries with >= 262144 candidates each take by
 * default -- an exhaustive quota = N run (lopq/lopq/search.py:128-133 consumes whole cells until the quota), or any quota over
 * cells of hundreds of thousands of codes: a sampled threshold per query, one pass over the codes at the HBM rate that lists the
 * few thousand candidates under it, exact float64 keys and ranking of the list, and a proof that the list holds the true top
 * `limit`; a query whose proof fails is answered again by the generic path).
 * All routes produce identical results; the switch exists so that tests can prove it. */
int cis_index_set_scan_mode(cis_index* ix, int mode);
/* counters[0] = batches the HBM-streaming route served, counters[1] = batches it handed back to the generic path (failed proof or
 * overflowed candidate list). */
int cis_index_stream_counters(cis_index* ix, int64_t counters[2]);
int cis_index_read_profile(cis_index* ix, double ms[5], int64_t* launches);

/* ---- CNN descriptors: replaces the caffe forward behind SentiBankPyCaffeImgFeaturizer.featurize -----
 * (cufacesearch/cufacesearch/featurizer/sbpycaffe_img_featurizer.py:137-154; network
 * cufacesearch/cufacesearch/featurizer/data/pycaffe_sentibank.prototxt:1-212).
 * arch CIS_CNN_SENTIBANK: tensors = {conv1_w, conv1_b, ..., conv5_w, conv5_b, fc6_w, fc6_b, fc7_w, fc7_b}
 * (14 float32 arrays in caffe layout: conv OIHW with I = in_channels / group, fc [out][in] over the
 * CHW-flattened blob).  forward: nchw [n][3][227][227] float32 exactly as preprocess_img (:113-134)
 * produces it (BGR, mean subtracted) -> feats [n][4096] float32 = blobs['fc7'] after the in-place ReLU. */
#define CIS_CNN_SENTIBANK 1
/* arch CIS_CNN_DLIB_RESNET: dlib's face-descriptor network behind DLibFeaturizer.featurize
 * (cufacesearch/cufacesearch/featurizer/dlib_featurizer.py:86-105, compute_face_descriptor :105).  The network
 * (dlib's anet_type, 29 convolutions) is third-party and not in the reference tree: oracle/dlib_oracle.py restates it.
 * tensors (117) = conv0 {w OIHW, b, affine gamma, affine beta}, then for each of the 14 residual blocks
 * {w, b, gamma, beta} of its first and of its second 3x3 convolution, then fc [128][256] (no bias).
 * forward input: aligned face chips [n][150][150][3] float32 RGB 0..255 (channels last) -> feats [n][128]. */
#define CIS_CNN_DLIB_RESNET 2
int cis_cnn_create(cis_cnn** out, int arch, const float* const* tensors, int n_tensors);
void cis_cnn_destroy(cis_cnn* c);
int cis_cnn_feat_dim(int arch);
int cis_cnn_forward(cis_cnn* c, const float* nchw, int n, float* feats);
int cis_cnn_forward_dev(cis_cnn* c, const float* d_nchw, int n, float* d_feats, void* stream);
/* A second handle on the same network (round 5): shares the weights of `base` (which must outlive it; a view whose base was destroyed
 * answers CIS_EINVAL), owns its workspaces, streams and events.  Several BATCHES in flight -- one per handle, each on its own stream --
 * fill each other's workgroup rounds: consecutive forwards are in different layers at any time.  From the first view on, base and views
 * run a batch as ONE chain (the two-part forward of the dlib net is for a single handle).  Destroy with cis_cnn_destroy. */
int cis_cnn_create_view(cis_cnn** out, cis_cnn* base);

/* ---- aligned face chips: the image side of dlib's compute_face_descriptor(img, shape) -------------------------------------
 * (cufacesearch/cufacesearch/featurizer/dlib_featurizer.py:103-105).  The caller supplies the 68 landmarks (the shape predictor is
 * not on this path); columbiaimagesearch_amd/featurizer/face_chip.py turns them into dlib's chip_details (get_face_chip_details(shape,
 * 150, 0.25): similarity transform onto the mean face) and into the affine map chip pixel -> source pixel of extract_image_chips.
 * cis_pyramid_down2_dev: dlib's pyramid_down<2> on an RGB uint8 image [nr][nc][3] -> [(nr-3)/2][(nc-3)/2][3] (d_tmp: nr*((nc-3)/2)*3 ints).
 * cis_extract_chips_dev: n chips [n][size][size][3] float32 0..255 (the input of cis_cnn_forward, CIS_CNN_DLIB_RESNET) out of one image
 * or pyramid level by dlib's interpolate_bilinear; d_maps [n][6] float64: source (x, y) = (m0 + m1 c + m2 r, m3 + m4 c + m5 r). */
int cis_pyramid_down2_dev(const uint8_t* d_img, int nr, int nc, uint8_t* d_out, int* d_tmp, void* stream);
int cis_extract_chips_dev(const uint8_t* d_img, int nr, int nc, const double* d_maps, int n, int size, float* d_chips, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CIS_HIP_H */
