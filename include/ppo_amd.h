/*
 * ppo_amd.h — C ABI of libppo_amd.so, the MI355X (gfx950) hot path of the PPO
 * trainer.  Plain pointers and sizes only; every pointer is a DEVICE pointer
 * unless its comment says host.  No entry point allocates, synchronises the
 * device or touches the host copy of any buffer; all work is enqueued on
 * `stream` (a hipStream_t passed as void*, NULL = the default stream) so calls
 * can be captured into a hipGraph.
 *
 * The reference (dremovd/PPO) is 100 % Python and has no FFI; each entry point
 * below replaces the Python function cited beside it.  INTEGRATION.md shows
 * the ctypes binding a maintainer of the reference would add.
 *
 * Return value: 0 on success, <0 on error (PPO_E_*); ppo_last_error() gives a
 * thread-local human-readable message.  Nothing is written on error.
 */
#ifndef PPO_AMD_H
#define PPO_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PPO_OK 0
#define PPO_E_INVALID (-1) /* bad shape / null pointer / unsupported kind */
#define PPO_E_HIP (-2)     /* a HIP runtime call failed */
#define PPO_E_ALIGN (-3)   /* pointer or leading dimension not aligned as documented */

/* dtype of the `terminals` / `dones` operand.  The reference's arithmetic
 * depends on it (NumPy promotion, rl/returns.py:24-28): bool terminals make
 * the recurrence run in float64 with only the stored result rounded to f32;
 * float32 terminals (or None) keep everything in float32. */
#define PPO_TERM_NONE 0 /* terminals == NULL: no episode ends inside the rollout */
#define PPO_TERM_U8 1   /* uint8 0/1 (NumPy bool)  -> float64 carry */
#define PPO_TERM_F32 2  /* float32 0.0/1.0         -> float32 carry */

/* which scan kernel to run */
#define PPO_SCAN_AUTO 0
#define PPO_SCAN_COLUMNS 1 /* one thread per 4 env columns, serial over time; bit-exact
                              with the reference's loop order; for wide batches */
#define PPO_SCAN_TILES 2   /* time axis split across the waves of a workgroup and
                              recombined through LDS in float64; for narrow batches */

int ppo_version(void);
const char *ppo_last_error(void);

/*
 * Fused GAE advantages + lambda-returns over a time-major rollout.
 * Replaces rl.returns.gae (rl/returns.py:7-29) and rl.returns.td_lambda
 * (rl/returns.py:58-67) as called by Runner.calculate_returns
 * (rl/rollout.py:1207-1223):
 *
 *   delta_t = r_t + gamma * V_{t+1} * (1 - term_t) - V_t        V_N = final_value
 *   adv_t   = delta_t + gamma*lam_adv * (1 - term_t) * adv_{t+1}         adv_N = 0
 *   g_t     = delta_t + gamma*lam_ret * (1 - term_t) * g_{t+1}           g_N   = 0
 *   ret_t   = f32(g_t) + V_t
 *
 * rewards, values, terminals, adv_out, ret_out: [N, ld] row-major, A <= ld
 * columns used (time-major, env index contiguous).  final_value: [A].
 * adv_out and ret_out may each be NULL (then that output is skipped).
 * gamma / lam_* are the python floats of the reference (double), because the
 * product gamma*lam is formed in double there.
 *
 * Algorithmic HBM traffic: 9 B read + 8 B written per (t, env) element.
 */
int ppo_gae_scan_f32(const float *rewards, const float *values, const float *final_value,
                     const void *terminals, int terminal_kind,
                     float *adv_out, float *ret_out,
                     int N, int A, int64_t ld,
                     double gamma, double lam_adv, double lam_ret,
                     int regime, void *stream);

/*
 * Discounted bootstrapped returns.  Replaces
 * rl.returns.calculate_bootstrapped_returns (rl/returns.py:32-55):
 *   G_t = r_t + G_{t+1} * gamma_t * (1 - done_t),  G_N = final_value.
 * gamma_arr: optional [N, ld] f32 per-element discount (NULL -> scalar gamma,
 * rounded to f32 as the reference does).  done_kind: PPO_TERM_U8 or PPO_TERM_F32.
 */
int ppo_bootstrapped_returns_f32(const float *rewards, const void *dones, int done_kind,
                                 const float *final_value, const float *gamma_arr, double gamma,
                                 float *out, int N, int A, int64_t ld, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PPO_AMD_H */
