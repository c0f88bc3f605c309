/*
 * bn_mi355x.h -- C ABI of the MI355X-native inference engine (libbn_mi355x.so).
 *
 * The reference (godai0519/BayesianNetwork) has no FFI layer: its boundary is the C++ class
 * surface of bayesian/inference.  These entry points are what a binding underneath that
 * surface needs, and each one names the reference interface it replaces.  The header-only
 * C++14 drop-in classes that sit on top are in include/bayesian/inference/ (same names and
 * signatures as the reference's); INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - Plain C types only; no exceptions cross the boundary.  Every int-returning function
 *     returns BN_OK (0) or a negative bn_status; bn_last_error() gives the message of the
 *     calling thread's last failure.
 *   - The caller keeps ownership of every array it passes; the engine copies what it needs.
 *   - An engine handle is NOT thread-safe (the reference's functors are not either:
 *     belief_propagation.hpp:320-333 holds mutable scratch state).
 *   - Node identity is the position in graph_t::vertex_list() (graph.hpp:214); parents are
 *     ascending (graph.hpp:389-402); the CPT of a node is row-major with the first parent most
 *     significant and the node's own state fastest (belief_propagation.hpp:269-295).
 */
#ifndef BN_MI355X_H
#define BN_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum bn_status {
    BN_OK = 0,
    BN_ERR_ARG = -1,        /* malformed model / evidence / argument            */
    BN_ERR_HIP = -2,        /* a HIP runtime call failed                         */
    BN_ERR_NO_DEVICE = -3,  /* no gfx950 device visible                          */
    BN_ERR_ALLOC = -4,      /* host allocation failed                            */
    BN_ERR_COMM = -5,       /* RCCL call failed / communicator not initialised   */
    BN_ERR_STATE = -6       /* call not valid in the engine's current state      */
} bn_status;

#define BN_MAX_PARENTS 16
#define BN_MAX_BATCH_SETS 256 /* evidence sets per bn_bp_run_batch call */
#define BN_DEVICE_HOST_ONLY (-2) /* build the layout plan only; no HIP call is made */
#define BN_DEVICE_CURRENT (-1)

/*
 * Flat model = graph_t + every vertex_t::cpt, flattened once.
 * Replaces: graph_t const& taken by belief_propagation::belief_propagation
 * (belief_propagation.hpp:16) and likelihood_weighting::likelihood_weighting
 * (likelihood_weighting.hpp:20), plus cpt_t lookups (graph.hpp:117).
 */
typedef struct bn_model_desc {
    int32_t n_nodes;
    const int32_t *k;       /* [n]    vertex_t::selectable_num                         */
    const int32_t *in_ptr;  /* [n+1]  CSR over parents                                 */
    const int32_t *in_idx;  /* [E]    parents, strictly ascending per node             */
    const int64_t *cpt_off; /* [n+1]  prefix sums of k[v] * prod k[parents]            */
    const double *cpt;      /* flat CPTs, reference row order                          */
    int32_t device;         /* HIP ordinal, BN_DEVICE_CURRENT or BN_DEVICE_HOST_ONLY   */
    int32_t lanes_per_node; /* 0 = automatic: small networks get a layout that shortens the   */
                            /*     latency of ONE query (more, lighter wavefronts: any-arity */
                            /*     tiles for nodes with many children, the wide lane-group   */
                            /*     split of 3 / 4 below);                                    */
                            /* 1 = one lane per node everywhere (no lane groups, no          */
                            /*     wavefront-per-node variant: A/B tests);                   */
                            /* 2 = dense: fewest wavefronts, for throughput (what the engine */
                            /*     builds internally for bn_bp_run_batch on such networks);  */
                            /* 3, 4 = 0, 2 plus the wide lane-group split (k = 4 with 3 or 4 */
                            /*     parents: 16 table entries per lane instead of 64) on small*/
                            /*     networks: ~10 % less latency per query, 4x the wavefronts;*/
                            /*     marginals agree with 0 / 2 to rounding, not bit for bit   */
} bn_model_desc;

typedef struct bn_engine bn_engine;

/* Validate the model, build the device layout, upload it.  Replaces the functor constructors. */
int bn_create(const bn_model_desc *desc, bn_engine **out);
void bn_destroy(bn_engine *eng);
/*
 * New CPT values on the SAME structure.  The reference reads node->cpt on every call (belief_propagation.hpp:61,186,252;
 * likelihood_weighting.hpp:148-158), so a table edited or re-fitted (sampler::make_cpt) after a functor was built is seen by
 * its next call; an engine holds device images made at bn_create and sees new values only through this call.
 * cpt [n_entries] is the whole flat array in bn_model_desc.cpt's layout; n_entries must equal the model's cpt_off[n].
 * Costs one pass over the tables on the host and one host-to-device copy per image; every later bn_bp_run* / bn_lw_run /
 * bn_rs_run uses the new values.  Staged evidence stays in force.
 */
int bn_reload_cpt(bn_engine *eng, const double *cpt, int64_t n_entries);

/*
 * Multi-GPU: one process per GPU, each creating shard `rank` of `nranks` from the SAME global
 * model.  owner[v] in [0,nranks) is the edge-cut partition (NULL = contiguous ranges balanced by
 * CPT bytes: row stripes on a row-major grid).  The pi-message of an edge is computed by the
 * parent's owner, the lambda-message by the child's owner; cut edges are exchanged by ONE
 * in-place RCCL all-gather per sweep, which also carries the residual so that every rank stops on
 * the same sweep.  Bootstrap: rank 0 calls bn_comm_unique_id, ships the 128 bytes to the other
 * ranks by any means (bench.py: torch.distributed broadcast), every rank calls bn_comm_init.
 * Beliefs of a sharded engine are node-major over ALL nodes with zeros for nodes of other ranks
 * (summing the ranks' arrays gives the global result).
 */
int bn_create_sharded(const bn_model_desc *desc, int32_t rank, int32_t nranks, const int32_t *owner,
                      bn_engine **out);
/*
 * Halo exchange INSIDE the resident kernel (no collective per sweep): every rank exports a blob -- handles of its
 * record buffers and sync block, and which of its tiles hold the nodes on cut edges -- the caller ships the blobs by
 * any means (bench.py / tests: torch.distributed all_gather_object, a pipe), and every rank imports all of them
 * (blobs[r] = rank r's, its own included).  After that a run of the sharded engine is ONE launch per rank: a tile
 * stores the message halves it produces for a cut edge into the peer's exchange region as well as its own
 * (peer-mapped memory: hipIpc handles between processes, plain pointers inside one; system-scope write-through
 * stores over xGMI), tiles wait for their neighbour tiles' generation granules wherever those live, and each rank's
 * service block hands its residual to every rank, so all ranks stop on the same sweep (belief_propagation.hpp:
 * 105-147).  Networks whose tiles the resident kernel does not cover (any rank) stay on bn_comm_init + per-sweep
 * launches; bn_get_info("shard_flow") tells.  Every rank must then run the same sequence of bn_bp_run* calls.
 */
int64_t bn_peer_blob_size(bn_engine *eng);
int bn_peer_export(bn_engine *eng, void *blob, int64_t cap);
int bn_peer_import(bn_engine *eng, const void *const *blobs, const int64_t *sizes, int32_t n);
int bn_comm_unique_id(void *id_out128);
int bn_comm_init(bn_engine *eng, const void *id128);
const char *bn_last_error(void);
const char *bn_version(void);

/*
 * Loopy belief propagation to convergence.
 * Replaces: belief_propagation::operator()(precondition, epsilon) (belief_propagation.hpp:31-159).
 *   ev_node[ne], ev_off[ne+1], ev_val[ev_off[ne]] : evidence vectors (1 x k[v] each, used as both
 *       pi and lambda, :68-73); ne may be 0 (the by-pass overload, :24-28).
 *   eps        : strict '<' on the maximum absolute message change (:147)
 *   max_sweeps : 0 = unbounded like the reference
 *   beliefs_out: [sum k] node-major, normalize(pi % lambda) (:151-158); host memory
 *   sweeps_out / residual_out : iterations executed and the last maximum_difference (optional)
 */
int bn_bp_run(bn_engine *eng, int32_t ne, const int32_t *ev_node, const int32_t *ev_off,
              const double *ev_val, double eps, int32_t max_sweeps, double *beliefs_out,
              int32_t *sweeps_out, double *residual_out);

/*
 * The same in two steps, for callers that keep inputs and outputs resident in HBM (bench.py times
 * this path): bn_bp_set_evidence validates, uploads and applies an evidence set once (it stays in
 * force until the next call); bn_bp_run_device runs belief propagation on it -- any number of
 * times, each run starts directly with its first sweep -- and leaves the beliefs in device
 * memory (bn_bp_beliefs_device, node-major [sum k]); only the sweep count and the last
 * maximum_difference come back.  bn_bp_run == set_evidence + run_device + copy_beliefs.
 */
int bn_bp_set_evidence(bn_engine *eng, int32_t ne, const int32_t *ev_node, const int32_t *ev_off,
                       const double *ev_val);
int bn_bp_run_device(bn_engine *eng, double eps, int32_t max_sweeps, int32_t *sweeps_out,
                     double *residual_out);
const double *bn_bp_beliefs_device(bn_engine *eng);
/* bn_bp_run with the marginals left in a page-locked host buffer owned by the engine (*beliefs_view, [sum k],
 * valid until the next run on this engine): evidence upload, run and the copy of the beliefs are queued back
 * to back and waited for ONCE, and a caller that unpacks the flat array anyway (the C++ functor building its
 * map of 1 x k matrices, belief_propagation.hpp:151-158) reads it in place. */
int bn_bp_run_view(bn_engine *eng, int32_t ne, const int32_t *ev_node, const int32_t *ev_off,
                   const double *ev_val, double eps, int32_t max_sweeps, const double **beliefs_view,
                   int32_t *sweeps_out, double *residual_out);
int bn_bp_copy_beliefs(bn_engine *eng, double *beliefs_out);

/*
 * Several evidence sets on ONE network in one call -- an extension beside the drop-in: the reference's
 * operator() (belief_propagation.hpp:31) takes one query at a time, and each set here gets exactly the
 * result that call would give it (same sweep count, same bits).  Networks the resident kernel covers
 * (bn_set_option "multisweep") run up to 4 sets per launch, walked round-robin: one resident CPT image
 * serves every set and each set's grid barrier completes while the others compute (more sets: consecutive
 * launches).  Every other network runs ALL sets in each per-sweep launch (one evidence set per blockIdx.y):
 * B queries share the launch, its latency and the CPT lines in the caches; a set that has converged drops
 * out of the following launches.  n_sets in 1..BN_MAX_BATCH_SETS.
 *   ne[n_sets]           : evidence nodes per set
 *   ev_node, ev_val      : the sets' arrays concatenated
 *   ev_off               : per set a block of ne[q] + 1 offsets STARTING AT 0, blocks concatenated
 *   beliefs_out          : [n_sets][sum k]; sweeps_out / residual_out : [n_sets]
 * bn_bp_set_evidence_batch + bn_bp_run_batch_device + bn_bp_copy_beliefs_batch == bn_bp_run_batch, with the
 * inputs and outputs left resident in HBM in between (bench.py times the middle call).
 */
int bn_bp_run_batch(bn_engine *eng, int32_t n_sets, const int32_t *ne, const int32_t *ev_node, const int32_t *ev_off,
                    const double *ev_val, double eps, int32_t max_sweeps, double *beliefs_out, int32_t *sweeps_out,
                    double *residual_out);
int bn_bp_set_evidence_batch(bn_engine *eng, int32_t n_sets, const int32_t *ne, const int32_t *ev_node,
                             const int32_t *ev_off, const double *ev_val);
int bn_bp_run_batch_device(bn_engine *eng, double eps, int32_t max_sweeps, int32_t *sweeps_out, double *residual_out);
int bn_bp_copy_beliefs_batch(bn_engine *eng, double *beliefs_out);
int bn_bp_residual_history_batch(bn_engine *eng, int32_t set, double *out, int32_t cap);

/* Diagnostics of the last bn_bp_run*: per-sweep maximum_difference (returns the count written),
 * and the final pi / lambda messages in CSR edge order (each sum_e k[parent(e)] doubles). */
int bn_bp_residual_history(bn_engine *eng, double *out, int32_t cap);
int bn_bp_messages(bn_engine *eng, double *pi_msg_out, double *lambda_msg_out);

/* Options: "timing" 1/0 -- HIP events on the engine's stream around every batch of sweep launches
 * (bn_bp_stats.sweep_kernel_ms).  Default 0 (BN_TIMING=1 in the environment turns it on): an event
 * record between two launches opens a bubble of several microseconds in the queue, so a timed run is
 * slower than an untimed one; bn_bp_stats.sweep_devclock_ms -- the device's 100 MHz clock read by the
 * kernels themselves at the first sweep's start and the last sweep's end -- costs nothing and is always on.
 * "overlap" 1/0 -- sharded runs: launch the interior tiles of a sweep while the previous sweep's
 *   all-gather is in flight on a second stream (default 1; BN_OVERLAP=0), or kernel and collective
 *   back to back on one stream.
 * "multisweep" 0/1/2 -- the one-launch path (BN_MULTISWEEP in the environment sets the default, 1):
 *   networks of one-lane tiles (uniform arity 2..4, <= 2 parents, <= 8 children per node) that fit the chip
 *   can run the whole run in ONE launch with CPTs, references and node vectors resident in registers / LDS
 *   and a grid barrier per sweep.  0 = always one launch per sweep; 1 = that path where it was measured
 *   faster (one-block networks, networks of >= 600 tiles); 2 = wherever eligible (tests, experiments).
 *   Results are bit-identical on either path.
 * "small" 0/1/2 -- SMALL networks (the state fits one CU's LDS: up to a few thousand CPT entries, <= 8 parents per
 *   node -- ALARM-sized): the whole run in ONE workgroup with messages, node vectors and staged terms in LDS, one
 *   work item per CPT entry / per message element instead of one wavefront per handful of nodes; sums and
 *   products in the reference's order for any table size, i.e. bit-identical to the CPU restatement.  0 = never,
 *   1 = where eligible and not measured slower than the resident tiles (default: everything eligible except long chains /
 *   trees and two-round networks the resident kernel runs in one block; "multisweep" 0 also turns it off), 2 = wherever
 *   eligible.  bn_bp_run_batch on such a network runs one workgroup per evidence set, all sets in one launch.
 *   bn_get_info "small_eligible".
 * "mid" 0/1/2 -- MID-SIZE networks (beyond one workgroup's LDS, up to 224 workgroups' worth: a few hundred to ten thousand nodes
 *   of mixed arity with <= 8 parents): the same items as "small", spread over several workgroups by node ranges, state in
 *   device memory, a grid barrier per iteration, one launch per run; bit-identical to the CPU restatement as well.  0 = never,
 *   1 = where eligible and not measured slower than the resident tiles (default: everything the resident tiles do not cover,
 *   and k = 4 networks with two parents per node), 2 = wherever eligible.  bn_bp_last_path = 4.
 *   Batches run as many sets per launch as fit the chip.  bn_get_info "mid_eligible", "mid_parts", "mid_aborts".
 * "flow" 1/0 -- resident path, one evidence set, more than one tile block (BN_RESIDENT_FLOW sets the default, 0):
 *   1 = dataflow form: a tile waits for the tiles it exchanges messages with instead of for a grid barrier, and
 *   the stop decision lags one iteration behind; 0 = grid barrier per sweep.  Same bits either way.
 * "direct" 1/0 -- resident path, grid-barrier form, one evidence set (BN_RESIDENT_DIRECT sets the default, 1): 1 = every tile block
 *   reads all blocks' arrival words itself and takes the stop decision (one hand-off per barrier); 0 = a service block collects
 *   them and publishes the decision (two).  Same bits.
 * "poll_sleep" n -- dataflow form: pause between two polls of a waiting tile, n x 512 cycles (default 2).
 * "beliefs_direct" 1/0 -- bn_bp_run_view: the kernels write the marginals straight into the engine's mapped host
 *   buffer (default 1, outputs up to 16 MB) instead of a copy command queued behind the run.
 * "dag" 0/1/2 -- networks whose nodes all have arity <= 4 and at most 5 parents (BASELINE configs[1], the 10 k-node random DAG; arities 2
 *   and 3 are padded to 4 with zeros, which leaves the real entries' bits alone): one
 *   launch per run with every CPT entry resident in a register; a node's child role (pi(v), lambda-messages: one wavefront of
 *   nodes / lane groups per tile) and parent role (lambda(v), pi-messages: one lane per message) run on different waves, the state
 *   lives in device memory in CSR edge order, one grid barrier per iteration.  Nodes with <= 2 parents keep the reference's
 *   operation order; with >= 3 parents the contraction is factored (sums over the two trailing parents shared by all outputs):
 *   results agree with the reference to rounding (<= 1e-12; its own products over >= 3 parents are unordered,
 *   belief_propagation.hpp:253).  Networks beyond one tile per wave run the same code walking several tiles per wave ("stream"
 *   form).  0 = never, 1 = where measured faster (default: networks with 3-5-parent nodes, and networks of <= 2-parent nodes that fit
 *   the chip at one tile per wave -- unless the one-workgroup path takes the network, or less than a quarter of the padded tables
 *   is real: binary networks of <= 2-parent nodes), 2 = wherever eligible.
 *   bn_bp_run_batch on such a network: up to 16 evidence sets share a launch and its CPT registers, taking turns inside an iteration;
 *   every set keeps the sweep count and the bits of its single run.
 *   bn_get_info "dag_eligible", "dag_blocks", "dag_tiles", "dag_stream", "dag_aborts".
 * "dagflow" 1 -- single queries on that path run in its DATAFLOW form where the plan has one (one tile per wave, more than one block, at
 *   most 64 neighbour tiles per tile): no grid barrier -- a tile starts its next iteration when the tiles it exchanges messages with
 *   have finished the previous one, one more block takes the stop decision one iteration behind, the one speculative iteration writes
 *   the other buffer.  Same sweep counts and bits as the barrier form.  0 (default): the barrier form -- measured faster on every
 *   network tried but a 64 x 64 grid (EXPERIMENTS.md R6.2).  bn_get_info "dag_flow_eligible" (known once the path has run or its plan
 *   was asked for), "dag_flow_max_nbr", "last_dag_flow".
 * "autotune" 1 -- the NEXT run first times every execution path the engine is eligible for on the evidence in force (one warm-up and
 *   two timed runs of 6 sweeps each, host wall clock) and keeps the fastest for all later runs: the built-in choice between the
 *   paths rests on thresholds measured on a handful of networks on one pool of machines.  The choice is written into the options
 *   above, which can still be set afterwards; bn_get_info "autotuned" (0 / 1) and "autotuned_path" tell.  0 clears a pending
 *   request.  Single-rank engines only.  Exchanging paths may change the last bits of networks with >= 3-parent nodes.
 * bn_bp_last_path: 0 = one launch per sweep, 2 = resident tiles (one launch per run), 3 = one workgroup, state in LDS
 *   (small networks, one launch per run), 4 = the same items over several workgroups (mid-size networks), 5 = register-resident
 *   child tiles + parent items (networks of arity <= 4 with <= 5 parents). */
int bn_set_option(bn_engine *eng, const char *name, int32_t value);
int bn_bp_last_path(bn_engine *eng);
/* Named integer properties (tests, tools): "resident_eligible", "flow_eligible", "last_flow" (1: the last run
 * used the dataflow form), "nbr_max", "nbr_chunks", "resident_blocks", "resident_aborts", "shard_flow" (in-kernel
 * exchange set up), "n_boundary_nodes", "rccl_ranks" (what the RCCL communicator of a sharded engine reports; 0: none), "small_eligible", "small_waves", "small_lds_bytes", "mid_eligible", "mid_parts", "mid_aborts", "dag_eligible", "dag_blocks", "dag_tiles", "dag_stream", "dag_aborts", "autotuned", "autotuned_path", "create_us_plan" / "create_us_small" / "create_us_mid" / "create_us_dag" / "create_us_device" (microseconds bn_create spent on the host plans -- tile layout, one-workgroup, several-workgroup, register-resident DAG -- and on the device side: allocations + uploads; bn_create builds the plan and image of the register-resident DAG path only where the defaults pick that path, elsewhere its image is filled and uploaded by the first run that wants it -- "dag" 2, "autotune", a batch), "lw_small" (1 once a sampler call has run: the straight-line sampling kernel for networks whose every node has <= 4 parents, <= 256 CPT rows and <= 4 states is in use); unknown name: BN_ERR_ARG.
 * When a one-launch path gives up a bounded wait (its workgroups were not all on the chip: another engine, stream or process uses the
 * GPU) the run is repeated on a slower path; the first such event of an engine prints ONE line on stderr, all are counted. */
int64_t bn_get_info(bn_engine *eng, const char *name);

/* Single steps of a run (tests / diagnostics): begin, one sweep (without exchange), finish.
 * bn_debug_allgather emulates the exchange between n shard engines living on ONE device. */
int bn_bp_step_begin(bn_engine *eng);
int bn_bp_step_sweep(bn_engine *eng, int32_t sweep, double eps);
/* One sweep in the two launches of a sharded run with the exchange overlapped: part 1 = the interior
 * tiles (they read nothing the previous sweep's all-gather delivers, so the engine launches them while
 * that collective is in flight), part 2 = the tiles that touch a cut edge + the residual bookkeeping
 * (launched once the collective has landed); part 0 = both in one launch (bn_bp_step_sweep). */
int bn_bp_step_sweep_part(bn_engine *eng, int32_t sweep, double eps, int32_t part);
int bn_bp_step_finish(bn_engine *eng, int32_t launched, int32_t final_batch, double eps,
                      int32_t *done_out, int32_t *sweeps_out, double *residual_out);
int bn_debug_allgather(bn_engine **engs, int32_t n, int32_t sweep);
/* The achievable HBM rate of a device, measured by the library's own streaming kernels (csrc/bn_stream.hip; 16 bytes per lane,
 * non-temporal loads and stores, HIP events on a stream of its own): mode 0 = copy (bytes read + bytes written), mode 1 = triad
 * (dst = a + s * b: 2 reads + 1 write); `bytes` = size of ONE array (take it well beyond the 256 MiB Infinity Cache); best of `reps`.
 * The yardstick SURVEY.md 8(d) asks for beside the nominal 8 TB/s; the reference has no counterpart.  device = HIP ordinal or
 * BN_DEVICE_CURRENT.  *gbs_out in GB/s. */
int bn_debug_stream(int32_t device, int32_t mode, int64_t bytes, int32_t reps, double *gbs_out);

typedef struct bn_bp_stats {
    int32_t sweeps;            /* iterations of the last run                                */
    int32_t sweep_launches;    /* sweep kernels launched (>= sweeps; extras exit at once)    */
    float sweep_kernel_ms;     /* HIP-event time over all sweep launches of the last run ("timing" on) */
    float total_ms;            /* host wall time of the last bn_bp_run_device call           */
    int64_t algorithmic_bytes_per_sweep; /* SURVEY.md 8(d) formula                           */
    int64_t layout_bytes_per_sweep;      /* bytes the sweep kernel actually requests         */
    int64_t messages_per_sweep;          /* 2E                                              */
    float sweep_devclock_ms;   /* device clock: first sweep's start -> last executed sweep's end */
    int32_t resident_aborts;   /* resident launches that gave up a bounded wait, over the engine's life: each sends
                                  the following runs down the per-sweep launches for a while (8, 16, ... runs) */
} bn_bp_stats;
int bn_bp_last_stats(bn_engine *eng, bn_bp_stats *out);

/*
 * Likelihood weighting.
 * Replaces: likelihood_weighting::operator()(evidence, sample_num) (likelihood_weighting.hpp:28-59).
 * Returns the UN-normalised weighted histogram [sum k] (node-major) of samples
 * [sample_begin, sample_begin + n_samples) so that several GPUs / calls can be summed; the
 * caller applies the reference's normalise rule (:197-221).  Every sample id owns one
 * xoshiro128++ stream seeded by a Philox4x32-10 block keyed by (`seed`, sample id) and advanced by one step per topological position
 * (oracle/lw_oracle.c states the mapping; the reference seeds an mt19937 from std::random_device, :224-244).
 */
int bn_lw_run(bn_engine *eng, int32_t ne, const int32_t *ev_node, const int32_t *ev_state,
              uint64_t sample_begin, uint64_t n_samples, uint64_t seed, double *hist_out);
/* The same over every rank of the communicator (bn_comm_init): the range [sample_begin,
 * sample_begin + n_samples_total) is split evenly, each GPU draws its share, one RCCL all-reduce
 * sums the histograms; every rank receives the total in hist_out. */
int bn_lw_run_allreduce(bn_engine *eng, int32_t ne, const int32_t *ev_node, const int32_t *ev_state,
                        uint64_t sample_begin, uint64_t n_samples_total, uint64_t seed, double *hist_out);
/*
 * Rejection (logic) sampling.
 * Replaces: rejection_sampling::operator()(condition, generate_sample_num)
 * (rejection_sampling.hpp:33-62): forward samples are drawn in index order until n_accept of them
 * agree with every (node, state) pair; counts_out [sum k] holds the state counts of exactly those
 * n_accept samples (the caller divides by the accepted count).  The reference loops forever when
 * the condition has probability zero; here at most max_draw samples are drawn and the numbers
 * actually drawn / accepted are reported.
 */
int bn_rs_run(bn_engine *eng, int32_t ne, const int32_t *ev_node, const int32_t *ev_state,
              uint64_t sample_begin, uint64_t n_accept, uint64_t max_draw, uint64_t seed,
              double *counts_out, uint64_t *drawn_out, uint64_t *accepted_out);
/*
 * Maximum-likelihood CPTs from a table of joint patterns.
 * Replaces: sampler::load_sample(table) + sampler::make_cpt(graph) (bayesian/sampler.hpp:29-37,
 * 81-163) -- the consumer of likelihood_weighting::make_samples' pattern table.
 *   patterns [n_patterns][n_nodes] : state of every node in each distinct pattern
 *   counts   [n_patterns]          : occurrences of each pattern
 *   structure                      : k / in_ptr / in_idx / cpt_off / device of the model; its `cpt`
 *                                    is not read (the graph has no CPTs yet) and may be NULL
 *   cpt_out                        : the fitted flat CPTs (layout of bn_model_desc.cpt); a row no
 *                                    pattern supports is uniform (:140-146)
 * An empty table (the reference's `return false`, :83) is BN_ERR_ARG.
 */
int bn_fit_cpt(const bn_model_desc *structure, int64_t n_patterns, const uint8_t *patterns,
               const uint64_t *counts, double *cpt_out);
/* Sampled states of the first `n` samples of the last bn_lw_run, sample-major [s][node]. */
int bn_lw_states(bn_engine *eng, uint64_t n, uint8_t *states_out, double *weights_out);

/* ---- layout introspection (host only; valid for BN_DEVICE_HOST_ONLY engines too) ---- */
typedef struct bn_layout_info {
    int32_t n_nodes, n_edges, n_classes, n_tiles;
    int32_t lanes_per_node_max;
    int64_t cpt_doubles, rec_doubles, node_doubles; /* striped device array sizes (one buffer) */
    int64_t algorithmic_bytes_per_sweep, layout_bytes_per_sweep, messages_per_sweep;
    int32_t rank, nranks, n_owned;  /* sharding: this rank's share */
    int32_t n_interior_tiles;       /* tiles [0, n_interior_tiles) touch no cut edge (== n_tiles on one rank) */
    int64_t n_cut_edges;            /* cut edges incident to this rank */
    int64_t segment_bytes;          /* all-gather payload per rank per sweep (residual slots included) */
    int64_t segment_used_bytes;     /* message halves this rank actually produces */
    int64_t exchange_base;          /* start of the exchange region in a record buffer, 16-byte units */
} bn_layout_info;
int bn_layout_get(bn_engine *eng, bn_layout_info *out);
/* per CSR edge: MsgRef {pi, lam} of bn_plan.hpp on this rank ({-1,0}: no owned endpoint) */
int bn_layout_edge_refs(bn_engine *eng, int32_t *pi_out, int32_t *lam_out);
/* node -> lane slot on this rank, -1 for nodes of other ranks, [n] */
int bn_layout_node_slots(bn_engine *eng, int32_t *slots_out);
/* node -> tile on this rank, -1 for nodes of other ranks, [n] */
int bn_layout_node_tiles(bn_engine *eng, int32_t *tiles_out);
/* per-class: kv, m, lanes_per_node, variant (0 = one-lane generic, 1 = register-resident template,
 * 2 = lane group (k = 4, 3-5 parents), 3 = any arities, a group of 8..64 lanes per node) */
/* dataflow tables: nbr_out [n_tiles * bn_get_info("nbr_chunks") * 64] neighbour slots (rank * 2048 + tile, -1 padded),
 * pub_out [n_tiles] bit q = the tile reports to rank q; either may be NULL */
int bn_layout_flow(bn_engine *eng, int32_t *nbr_out, uint32_t *pub_out);
int bn_layout_class(bn_engine *eng, int32_t cls, int32_t *kv, int32_t *m, int32_t *lanes_per_node,
                    int32_t *variant, int32_t *n_nodes);
/* The plan of the one-workgroup path for small networks (csrc/bn_small.hpp; tests emulate the kernel on it):
 * dims_out[12] = n, N (sum of arities), M (sum over edges of the parent's arity), S (CPT entries), T (staged terms),
 * TT (parent terms), CL (child-list entries), waves, re, rb, rc (rounds per item kind), mmax; the arrays (any may be
 * NULL) are sized from those: ent [re * 64 * waves][2], ent_cpt [re * 64 * waves], term [TT], clist [CL],
 * bslot [rb * 64 * waves][4], cslot [rc * 64 * waves][4], npi_init [N].  BN_ERR_STATE: the network is not eligible. */
int bn_small_plan_get(bn_engine *eng, int32_t *dims_out, uint32_t *ent, double *ent_cpt, uint32_t *term, uint16_t *clist,
                      uint32_t *bslot, uint32_t *cslot, double *npi_init);
/* ... of part `part` (0 .. bn_get_info "mid_parts" - 1) of the plan that spreads a mid-size network over several workgroups
 * (csrc/bn_mid.hip): same arrays; message and node-vector indices are global, staging places the part's own;
 * dims_out[14]: the twelve values above, then the part's node range [v0, v1). */
int bn_mid_plan_get(bn_engine *eng, int32_t part, int32_t *dims_out, uint32_t *ent, double *ent_cpt, uint32_t *term, uint16_t *clist,
                    uint32_t *bslot, uint32_t *cslot, double *npi_init);

/* The plan of the register-resident DAG path (csrc/bn_dag.hpp: networks of arity <= 4 with <= 5 parents per node; tests emulate the
 * kernel on it).  dims_out[8] = n, E, tiles, blocks, stream (1: some wave walks several tiles per iteration), child tiles, parent
 * tiles, doubles of the CPT image; the arrays (any may be NULL) are sized from those: tiles [tiles][8] (kind, active nodes / items,
 * first per-lane entry, first double2 of the CPT image, largest child count, 3 unused), slot_ptr [blocks * 8 + 1], cnode
 * [tiles * 64][2] (node, first in-edge), pitem [tiles * 64][4] (node, target out-edge or -1, first out-edge entry, child count |
 * target's rank << 16), oedge [max(E, 1)], cpt_img [dims_out[7]], npi_init [4 n].  BN_ERR_STATE: the network is not eligible. */
int bn_dag_plan_get(bn_engine *eng, int32_t *dims_out, int32_t *tiles, int32_t *slot_ptr, int32_t *cnode, int32_t *pitem,
                    int32_t *oedge, double *cpt_img, double *npi_init);

#ifdef __cplusplus
}
#endif
#endif /* BN_MI355X_H */
