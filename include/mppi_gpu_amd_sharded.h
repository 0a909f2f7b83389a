/*
 * mppi_gpu_amd_sharded.h -- C ABI of the single-process, multi-GPU MPPI controller.
 *
 * The reference has no multi-GPU path; its host loop (reference
 * src/main.cu:309-374) drives ONE `PointMassModel`.  This is the same controller surface
 * (reference include/point_mass.hpp:23-44, one function per public member) over the GPUs of one
 * node, for a C or C++ host that is ONE process: the K samples of a solve are split into
 * contiguous shards, one engine (include/mppi_gpu_amd.h: mppi_create_shard) per device, one host
 * worker thread per engine; noise comes from the Philox subsequences of the GLOBAL sample indices,
 * so the controls do not depend on the number of GPUs.  Per solve every shard contributes
 * T*A+2 floats [beta_g, S_g, N_g[T*A]]; how they travel is the TRANSPORT:
 *
 *   MPPI_XPORT_COLLECTIVE (default)  rank-local combine -> ncclAllGather (RCCL over xGMI, called
 *        natively: ncclCommInitAll once, one communicator per device) -> final combine.  What
 *        BASELINE.json's north_star names ("a single RCCL all-reduce ... on the per-action weighted
 *        sums and normaliser"; an all-gather + local combine here, because min and sum must both
 *        be carried in ONE round and the result must be bit-identical on every rank).
 *   MPPI_XPORT_DIRECT   no collective library on the data path: the rank-local combine kernel
 *        stores the partial straight into every peer's inbox (peer-mapped uncached device memory,
 *        8-byte {value, tag} words) and polls its own; rollout + ONE launch per solve, which rides
 *        in the next solve's rollout launch when solves follow each other (mppi_xchg_* of the
 *        single-engine ABI, same_process pointers + hipDeviceEnablePeerAccess).
 *   MPPI_XPORT_COPY     rank-local combine -> hipMemcpyPeerAsync of every partial -> final
 *        combine: needs neither RCCL nor peer stores; the rehearsal transport (RCCL refuses two
 *        ranks on one device) and the fallback of last resort.
 * All three run the same rank-local combine and the same final arithmetic: equal bits.
 *
 * Library: mppi_gpu_amd/lib/libmppi_gpu_amd_sharded.so (links libmppi_gpu_amd.so and librccl).
 * Conventions as in mppi_gpu_amd.h: host pointers, the reference's layouts (global sample order:
 * X[k][t][s], E[k][t][a] with k over ALL shards), 0 / negative MPPI_E* return codes,
 * mppi_sharded_last_error().  One host thread drives a controller at a time.  No CPU fallback.
 */
#ifndef MPPI_GPU_AMD_SHARDED_H_
#define MPPI_GPU_AMD_SHARDED_H_

#include "mppi_gpu_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mppi_sharded mppi_sharded;

enum { MPPI_XPORT_COLLECTIVE = 0, MPPI_XPORT_DIRECT = 1, MPPI_XPORT_COPY = 2 };

/* PointMassModel::PointMassModel over n_shards engines.  devices[i] = HIP device ordinal of shard
 * i (NULL: 0, 1, .., n_shards-1); n_shards = 0: one shard per visible device.  A device may be
 * named more than once (rehearsal on one GPU: transports DIRECT and COPY only).  nb_sim_global is
 * split into contiguous, balanced ranges (the first nb_sim_global % n_shards shards hold one
 * sample more).  reference src/point_mass.cu:19-106 */
int mppi_sharded_create(int nb_sim_global, int steps, float dt, int state_dim, int act_dim,
                        int verbose, int n_shards, const int* devices, int transport,
                        mppi_sharded** out);
/* reference src/point_mass.cu:108-127 */
void mppi_sharded_destroy(mppi_sharded* s);

/* memcpy_set_data / set_x / get_x / get_u: reference src/point_mass.cu:205-228, 482-486, 488-491
 * (every shard gets the same x0, U, goal, w; U is identical on all shards after every solve) */
int mppi_sharded_set_data(mppi_sharded* s, const float* x0, const float* u, const float* goal,
                          const float* w);
int mppi_sharded_set_x(mppi_sharded* s, const float* x0);
int mppi_sharded_get_x(mppi_sharded* s, float* x0);
int mppi_sharded_get_u(mppi_sharded* s, float* u);

/* get_act, reference src/point_mass.cu:129-203: one solve over all shards, blocking; every shard
 * arrives at the same action bit for bit (checked: MPPI_ESTATE if they differ). */
int mppi_sharded_get_act(mppi_sharded* s, float* next_act);
/* the same split in two: enqueue one solve on every shard / wait and read the action */
int mppi_sharded_solve_async(mppi_sharded* s);
int mppi_sharded_sync_act(mppi_sharded* s, float* next_act);

/* get_inf / memcpy_get_data, reference src/point_mass.cu:236-262, 230-234: per-sample outputs are
 * gathered from the shards in global sample order; beta, nabla, weights are the GLOBAL ones. */
int mppi_sharded_get_inf(mppi_sharded* s, float* x_all, float* u, float* noise, float* cost,
                         float* beta, float* nabla, float* weight);
int mppi_sharded_get_data(mppi_sharded* s, float* x_all, float* noise);

/* extensions, forwarded to every shard (see mppi_gpu_amd.h) */
int mppi_sharded_set_params(mppi_sharded* s, float lambda, const float* sigma, const float* inv_s);
int mppi_sharded_set_seed(mppi_sharded* s, unsigned long long seed);
int mppi_sharded_set_noise(mppi_sharded* s, const float* noise /* [K_global][T][A] or NULL */);
int mppi_sharded_set_action_limit(mppi_sharded* s, const float* max_a);
/* exchange time-out of the DIRECT transport (default 5 s) */
int mppi_sharded_set_timeout(mppi_sharded* s, double seconds);

/* introspection */
int mppi_sharded_n_shards(const mppi_sharded* s);
int mppi_sharded_transport(const mppi_sharded* s);
/* out = { k_begin, k_end, device } of shard i */
int mppi_sharded_shard_info(const mppi_sharded* s, int i, long long out[3]);
/* the engine of shard i (borrowed; for geometry / launch-count queries from the creating thread's
 * point of view -- call it only while no solve is in flight) */
mppi_engine* mppi_sharded_engine(mppi_sharded* s, int i);
const char* mppi_sharded_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* MPPI_GPU_AMD_SHARDED_H_ */
