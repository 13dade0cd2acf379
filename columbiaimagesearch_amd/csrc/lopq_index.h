// Internal view of the LOPQ index handle, shared by lopq_index.hip (storage: device-side insert) and lopq_search.hip
// (search pipeline).
//
// The index lives in HBM and only there.  Inserts -- from host arrays (cis_index_add) or from device arrays
// (cis_index_add_dev) -- are merged into the cell-contiguous arrays by kernels (lopq_index.hip); the host keeps no copy
// of codes or ids and no (cell, id) set.
#pragma once
#include "lopq_model.h"

// One cell-contiguous store: items of a cell in insertion order (= the reference's per-cell list order,
// lopq/lopq/search.py:359).  Two generations of every array: a merge reads the current one and writes the other.
struct CellStore {
    DevBuf codes[2];  // [n][M] uint8 (not allocated for the id-only store)
    DevBuf ids[2];    // [n] int64
    DevBuf loff[2];   // [ncells + 1] int64 starts of the cells' rooms, then [ncells] int64 ends of their USED parts (lend)
    DevBuf cmax;      // [ncells] int64: largest id stored in the cell (-1: empty) -- an id above it cannot be a duplicate
    int cur = 0;
    int64_t n = 0;    // items stored
    int64_t cap = 0;  // upper bound of the layout's extent (slots incl. the cells' slack; the true extent is loff[ncells] on the device)
    bool with_codes = true;
    bool init = false;
    void release() {
        for (int i = 0; i < 2; ++i) { codes[i].release(); ids[i].release(); loff[i].release(); }
        cmax.release();
    }
};

struct cis_index {
    cis_model* m = nullptr;
    int V = 0, M = 0;
    int64_t ncells = 0;
    int rank = 0, world = 1;
    std::vector<int32_t> owner;  // empty: cell % world
    DevBuf d_owner;              // [ncells] int32 when `owner` is set
    // HBM-resident index of THIS shard
    CellStore own;     // codes + ids of the cells this shard owns
    CellStore ghost;   // ids only, cells owned by OTHER shards: exists only when a sharded index is handed every item
                       // with dedup (cis_index_add on all ranks instead of the routed insert) -- what the duplicate
                       // test of those cells needs
    DevBuf d_gcount;   // [ncells] int64, all shards
    DevBuf d_plan_hint;  // [2][2] uint64 (base index; views read the base's): see k_plan_par
    bool had_plain_remote = false;  // items of other shards' cells were counted without (cell, id) bookkeeping
    int64_t nb_indexed = 0;
    bool stats_fresh = false;  // the last insert chunk already refreshed n_total / max_cell / nonempty_cells
    int64_t n_inplace = 0, n_rebuild = 0;  // insert calls (chunks) that were written in place / that rebuilt the layout
    // A VIEW (cis_index_create_view) shares the storage of `base` -- codes, ids, offsets, cell sizes -- and owns only its per-batch
    // workspaces, plan read-back words and counters: two batches can then be in flight at once, each on its own stream (the front end,
    // tables and slot kernels of one batch fill the tail of the other's scan and its merge).  A view is read-only and must not outlive
    // its base; inserts into the base must be ordered against the views' searches by the caller (events), like searches on the base.
    cis_index* base = nullptr;
    std::vector<cis_index*> views;  // base index: its live views (cis_index_destroy of the base orphans them instead of leaving them dangling)
    bool orphaned = false;          // view: its base was destroyed first -- every search through it fails with CIS_EINVAL
    const cis_index* st() const { return base ? base : this; }
    // views the search pipeline reads (current generation of `own`)
    const uint8_t* codes_ptr() const { const cis_index* s = st(); return s->own.codes[s->own.cur].as<uint8_t>(); }
    const int64_t* ids_ptr() const { const cis_index* s = st(); return s->own.ids[s->own.cur].as<int64_t>(); }
    const int64_t* loff_ptr() const { const cis_index* s = st(); return s->own.loff[s->own.cur].as<int64_t>(); }
    const int64_t* gcount_ptr() const { return st()->d_gcount.as<int64_t>(); }
    unsigned long long* plan_hint_ptr() const { return st()->d_plan_hint.as<unsigned long long>(); }
    void sync_from_base() {  // scalars of the storage the search path reads
        if (!base) return;
        ncells = base->ncells; rank = base->rank; world = base->world; nb_indexed = base->nb_indexed; n_local = base->n_local;
        n_total = base->n_total; max_cell = base->max_cell; nonempty_cells = base->nonempty_cells;
    }
    int64_t n_local = 0;
    // insert workspace
    DevBuf wi_key[2], wi_val[2], wi_hist, wi_sid, wi_acc, wi_apre, wi_tmp, wi_in_ids, wi_in_coarse, wi_in_fine, wi_scan, wi_cnt, wi_cnt2;
    DevBuf d_stats;              // statistics words of the last merge (see lopq_index.hip)
    int64_t* h_ins = nullptr;    // pinned host copy of them
    // per-batch workspace
    DevBuf w_slack;  // per work item: see k_merge_survivors
    DevBuf w_planfb, w_vis;  // k_plan_par: per-query fallback flags, visited (i, j) lists
    DevBuf w_tiles;          // tile sums of the candidate layout
    DevBuf w_xp, w_cd, w_order, w_sorted, w_plan, w_off, w_items, w_tabs, w_T, w_hits, w_hitn, w_part, w_q,
        w_oids, w_odists, w_onf, w_ovis, w_ocell, w_opos, w_order2, w_px, w_T32, w_grp, w_tord, w_y64, w_x64;
    int64_t stats[4] = {0, 0, 0, 0};
    // optional stage timing (hipEvents on the launch stream)
    bool force_exact_scan = false;  // tests: run every item through the float64 kernel
    bool force_scan2 = false;       // scan mode 2: the float32-prefilter kernel whatever the batch size
    bool force_scan3 = false;       // scan mode 3: the 16-bit fixed-point kernel whatever the batch size
    int force_two_pass = -1;        // scan mode 3: k_adc_scan3's streaming form, 4: its two-pass form (-1: by chunk length)
    int batch_hint = 0;             // sub-batch size that fitted the workspace budget after a retry (search_all)
    int64_t batch_hint_quota = -1;
    double retry_fraction = 0.5;
    int m16_holdoff = 0;            // batches the sampled scan at M = 16 stays off after a batch where its scale missed (search_batch)
    int64_t m16_backoffs = 0;
    int last_scan_kernel = 0;       // 0 none (all-candidates path), 1 float64 scan, 2 float32 prefilter, 3 16-bit fixed-point prefilter, 4 its sampled single-pass form (k_adc_scan4), 5 the HBM-streaming scan (k_adc_stream), 6 k_adc_scan5
    bool force_prefilter_scan = false;  // tests: the float32-prefilter kernel also for small batches
    hipStream_t h_stream = nullptr; // the host-pointer entry points' own stream (cis_index_search[_async])
    hipEvent_t h_ev_in = nullptr, h_ev_out = nullptr, h_ev_done = nullptr;  // copy-in landed / search done / copy-out landed
    bool h_pending = false;         // a batch of cis_index_search_async is in flight on it
    bool h_out_enqueued = false;    // ... and its copy-out is on the copy stream already (cis_host_pump_locked: lopq_search.hip)
    struct HostOut { int64_t* ids; double* dists; int32_t* n_found; int32_t* visited; int32_t* cells; uint32_t* pos; int nq, L; };
    HostOut h_out = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0};  // where its results go (copied out by cis_index_search_wait)
    bool force_scan5 = false;       // scan mode 7 (tests): k_adc_scan5 (one threshold per query, eight queries per slot) whatever the batch size
    DevBuf w_s5;                    // its per-batch buckets, counters and thresholds
    DevBuf w_bmin;                  // the streaming route's sample buckets [nq][STREAM_B]: "empty" between batches (k_stream_tau resets what it reads)
    bool force_stream = false;      // scan mode 6 (tests): the HBM-streaming route (lopq_stream.hip) whatever the batch looks like
    bool stream_off = false;        // internal: the batch is being answered again through the generic path after a failed proof
    int64_t stream_batches = 0, stream_fallbacks = 0;  // batches the streaming route served / that it handed back to the generic path
    int profiling = 0;  // 0 off, 1 events around the scan kernel only, 2 events around every stage
    int64_t* h_totals = nullptr;    // pinned, device-mapped: the plan totals land here without a copy
    int64_t* d_h_totals = nullptr;
    int64_t plan_seq = 0;           // sequence number of the last plan whose totals were requested
    int64_t stats_pending_seq = 0;  // != 0: the last batch did not read its totals back; last_stats waits for this plan
    int64_t n_total = 0, max_cell = 0, nonempty_cells = 0;  // over all shards (gcount): bounds for such batches
    struct ProfRec { hipEvent_t ev[6]; bool has_scan; };  // ev[5]: just before the scan kernel (after slot building)
    std::vector<ProfRec> prof;
    double prof_ms[5] = {0, 0, 0, 0, 0};
    int64_t prof_launches = 0;

    bool owns(int64_t cell) const {
        if (world <= 1) return true;
        if (!owner.empty()) return owner[cell] == rank;
        return (int)(cell % world) == rank;
    }
};

// Makes the device arrays exist (empty index: offsets all zero) -- called at the head of every search.
int cis_index_ready(cis_index* ix);

// lopq_search.hip: a handle that is being destroyed leaves the list of handles whose copy-out is still to be enqueued
void cis_host_forget(cis_index* ix);
