// engine.hpp -- interface between the C ABI layer (capi.cpp) and the device engine.
#pragma once
#include "mvx_internal.hpp"

namespace mvx {

int device_count();
int set_device(int dev);
void sync_stream();
int take_last_error(); // MVX_ENOMEM ... since the last call; clears it
int bind_thread();

// solve (glp_simplex)
int engine_simplex(mvx_prob *P, const mvx_smcp *parm);
int engine_simplex_batch(mvx_prob **probs, int count, const mvx_smcp *parm, int *rcs);

// tableau maintenance under model edits; all no-ops while !P->valid
void engine_apply_bounds(mvx_prob *P, int k, int type, double old_lb, double old_ub, double lb, double ub);
void engine_add_rows(mvx_prob *P, int first, int nrs); // P->m already updated
void engine_row_from_model(mvx_prob *P, int i);        // row i's auxiliary is basic: rebuild its tableau row
void engine_recompute_cost_row(mvx_prob *P);
void engine_invalidate(mvx_prob *P); // drop device state; next solve starts from the slack basis

void engine_copy(mvx_prob *dst, const mvx_prob *src); // device-to-device clone of the slab
void release_device(mvx_prob *P);

// host mirrors of beta / reduced costs / basis (cheap when already fresh)
void refresh_solution(const mvx_prob *P);
int engine_get_tableau(const mvx_prob *P, double *out);
int engine_get_row(const mvx_prob *P, int row, double *out); // out[0..n]

// GMI cuts of a solved node on the device (gmi.cpp:11-117); see engine.cpp
int engine_gmi_cuts(const mvx_prob *P, int mode, const int *cols, int count, double *vals, double *rhs, int *ok);
int engine_gmi_cuts_many(const mvx_prob *const *Ps, int mode, const int *cols, int count, double *vals, double *rhs, int *ok);

long long engine_pack_size(const mvx_prob *P, int m_base);
int engine_pack(const mvx_prob *P, int m_base, void *dev_buf);
int engine_unpack(mvx_prob *dst, const void *dev_buf);
void *engine_image_alloc(size_t bytes);
void engine_image_free(void *p);
void tuning(int tr, int hot, int nt);
void set_stall_limit(int limit);
int fcs_debug_stamps(unsigned long long *out);
void set_persist(int mode);
void set_chain(int len);
void set_cluster(int on);
void cluster_stats(long long *launches, long long *aborts);
void set_dual_chain(int len);
void set_refresh(int check_every, double tol);
double row_residual(const mvx_prob *P);
void persist_stats(long long *launches, long long *aborts);
void persist_cycles(unsigned long long *out5);
void set_batch_slots(int k);
void profile_enable(int on);
void profile_reset();
double profile_update_ms();
long long profile_update_launches();

} // namespace mvx
