// Synthetic vectorised environment stepped on host cores (no device code in this file).
//
// The benchmark workload of SURVEY.md §8(d) / BASELINE.md §3: observations are uint8 i.i.d.
// uniform[0,255] from a seeded counter-based generator, reward ~ N(0,1), done ~ Bernoulli(p),
// info = {time, ep_length, ep_score}; episodes auto-reset like gym's vector envs
// (rl/hybridVecEnv.py:24-46).  It stands where the reference has ALE / Procgen / MuJoCo worker
// processes (rl/hybridVecEnv.py:49-203): same call shape — step(actions[A]) fills obs, reward,
// done for all A envs — with the observation written straight into a caller-owned (pinned) host
// buffer so the trainer can issue one async H2D copy per step.  Action -1 skips an env
// (NullActionWrapper, rl/wrappers.py:1393-1418).
//
// A persistent pool of worker threads splits the envs; every value depends only on
// (seed, global env index, that env's step count), so results are independent of the thread count
// and of how envs are sharded over ranks.
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "common.h"

namespace {

inline uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// dst[8 i .. 8 i + 7] = mix64(key + (i + 1) * golden), i = 0 .. : the observation stream of one env step.  The words are
// independent, so the loop vectorises; the library is built once and runs on whatever host the GPU box has, hence one
// clone per vector ISA picked at load time (AVX-512DQ has the 64-bit multiply; AVX2 builds it from 32-bit ones).  Same
// integer arithmetic in every clone: the bytes do not depend on the host.
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
__attribute__((target_clones("avx512dq", "avx2", "default")))
#endif
void fill_words(uint8_t *__restrict__ dst, int64_t n_bytes, uint64_t key)
{
    const int64_t n8 = n_bytes / 8;
    for (int64_t i = 0; i < n8; ++i) {
        const uint64_t v = mix64(key + (uint64_t)(i + 1) * 0x9E3779B97F4A7C15ull);
        std::memcpy(dst + i * 8, &v, 8);
    }
    if (n_bytes % 8) {
        const uint64_t v = mix64(key + (uint64_t)(n8 + 1) * 0x9E3779B97F4A7C15ull);
        std::memcpy(dst + n8 * 8, &v, (size_t)(n_bytes % 8));
    }
}

struct SynthEnv {
    int n_envs;
    int64_t obs_bytes;
    uint64_t seed;
    double p_done;
    int64_t env_offset;  // global index of env 0 (data-parallel sharding)
    std::vector<int64_t> steps;     // per-env total step count (drives the generator)
    std::vector<int32_t> time;      // steps since episode start
    std::vector<float> score;       // undiscounted episode score
    // job description for the pool
    const int32_t *actions = nullptr;
    uint8_t *obs = nullptr;
    float *reward = nullptr;
    uint8_t *done = nullptr;
    int32_t *time_out = nullptr;
    float *ep_score_out = nullptr;
    int32_t *ep_len_out = nullptr;
    bool is_reset = false;
    // chunked step (ppo_synth_env_step_upload): envs are handed out in index order in blocks of `block`, and
    // chunk_left[c] counts the blocks of chunk c still running, so the caller can upload chunk c while later ones
    // are still being generated
    int block = 0, n_chunks = 0, blocks_per_chunk = 0;
    std::atomic<int> next_block{0};
    std::atomic<int> chunk_left[16];

    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_start, cv_done;
    uint64_t generation = 0;
    int pending = 0;
    bool stop = false;

    void fill_obs(int e)
    {
        // 8 bytes per mix of (seed, env, step, word)
        const uint64_t key = mix64(seed ^ mix64((uint64_t)(env_offset + e) * 0x9E3779B97F4A7C15ull + (uint64_t)steps[e]));
        fill_words(obs + (int64_t)e * obs_bytes, obs_bytes, key);
    }

    void run_range(int lo, int hi)
    {
        for (int e = lo; e < hi; ++e) {
            if (is_reset) {
                steps[e] = 0;
                time[e] = 0;
                score[e] = 0.f;
                fill_obs(e);
                continue;
            }
            if (actions && actions[e] < 0) {  // skipped env: nothing advances, obs unchanged
                if (reward) reward[e] = 0.f;
                if (done) done[e] = 0;
                if (time_out) time_out[e] = time[e];
                if (ep_score_out) ep_score_out[e] = score[e];
                if (ep_len_out) ep_len_out[e] = time[e];
                continue;
            }
            steps[e] += 1;
            const uint64_t k = mix64(seed * 0xD1342543DE82EF95ull + (uint64_t)(env_offset + e)) + (uint64_t)steps[e] * 3;
            const double u1 = ((double)(mix64(k) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
            const double u2 = ((double)(mix64(k + 1) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
            const double u3 = ((double)(mix64(k + 2) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
            const float r = (float)(std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2));
            const bool d = u3 < p_done;
            time[e] += 1;
            score[e] += r;
            if (reward) reward[e] = r;
            if (done) done[e] = d ? 1 : 0;
            if (time_out) time_out[e] = time[e];
            if (ep_score_out) ep_score_out[e] = score[e];
            if (ep_len_out) ep_len_out[e] = time[e];
            if (d) {  // auto-reset: the returned obs is the first obs of the next episode
                time[e] = 0;
                score[e] = 0.f;
            }
            fill_obs(e);
        }
    }

    void worker(int idx, int n_workers)
    {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_start.wait(lk, [&] { return stop || generation != seen; });
                if (stop) return;
                seen = generation;
            }
            if (block > 0) {
                // dynamic, in-order hand-out: the first chunk's envs are all done before most of the last chunk's start
                const int n_blocks = (n_envs + block - 1) / block;
                for (;;) {
                    const int b = next_block.fetch_add(1, std::memory_order_relaxed);
                    if (b >= n_blocks) break;
                    const int lo = b * block, hi = lo + block < n_envs ? lo + block : n_envs;
                    run_range(lo, hi);
                    chunk_left[b / blocks_per_chunk].fetch_sub(1, std::memory_order_release);
                }
            } else {
                const int per = (n_envs + n_workers - 1) / n_workers;
                const int lo = idx * per;
                const int hi = lo + per < n_envs ? lo + per : n_envs;
                if (lo < hi) run_range(lo, hi);
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                if (--pending == 0) cv_done.notify_all();
            }
        }
    }

    void dispatch()
    {
        if (workers.empty()) {
            run_range(0, n_envs);
            return;
        }
        std::unique_lock<std::mutex> lk(mu);
        pending = (int)workers.size();
        ++generation;
        cv_start.notify_all();
        cv_done.wait(lk, [&] { return pending == 0; });
    }

    void start()  // dispatch without waiting
    {
        std::unique_lock<std::mutex> lk(mu);
        pending = (int)workers.size();
        ++generation;
        cv_start.notify_all();
    }

    void join()
    {
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return pending == 0; });
    }
};

}  // namespace

extern "C" void *ppo_synth_env_create(int n_envs, int64_t obs_bytes, uint64_t seed, double p_done,
                                      int64_t env_offset, int n_threads)
{
    if (n_envs <= 0 || obs_bytes <= 0 || n_threads < 0) {
        ppo::fail(PPO_E_INVALID, "ppo_synth_env_create: bad arguments");
        return nullptr;
    }
    auto *e = new SynthEnv();
    e->n_envs = n_envs;
    e->obs_bytes = obs_bytes;
    e->seed = seed;
    e->p_done = p_done;
    e->env_offset = env_offset;
    e->steps.assign(n_envs, 0);
    e->time.assign(n_envs, 0);
    e->score.assign(n_envs, 0.f);
    if (n_threads > n_envs) n_threads = n_envs;
    for (int i = 0; i < n_threads; ++i) e->workers.emplace_back(&SynthEnv::worker, e, i, n_threads);
    return e;
}

extern "C" void ppo_synth_env_destroy(void *h)
{
    auto *e = static_cast<SynthEnv *>(h);
    if (!e) return;
    {
        std::lock_guard<std::mutex> lk(e->mu);
        e->stop = true;
    }
    e->cv_start.notify_all();
    for (auto &t : e->workers) t.join();
    delete e;
}

extern "C" int ppo_synth_env_reset(void *h, uint8_t *obs_out)
{
    auto *e = static_cast<SynthEnv *>(h);
    if (!e || !obs_out) return ppo::fail(PPO_E_INVALID, "ppo_synth_env_reset: null");
    e->obs = obs_out;
    e->is_reset = true;
    e->dispatch();
    e->is_reset = false;
    return PPO_OK;
}

extern "C" int ppo_synth_env_step(void *h, const int32_t *actions, uint8_t *obs_out, float *reward_out,
                                  uint8_t *done_out, int32_t *time_out, float *ep_score_out, int32_t *ep_len_out)
{
    auto *e = static_cast<SynthEnv *>(h);
    if (!e || !obs_out) return ppo::fail(PPO_E_INVALID, "ppo_synth_env_step: null");
    e->actions = actions;
    e->obs = obs_out;
    e->reward = reward_out;
    e->done = done_out;
    e->time_out = time_out;
    e->ep_score_out = ep_score_out;
    e->ep_len_out = ep_len_out;
    e->dispatch();
    return PPO_OK;
}

// Step all envs and upload their observations: the host-to-device copy of chunk c (1 / n_chunks of the envs, in index
// order) is queued on `stream` as soon as that chunk's observations exist, while the worker threads are still generating
// the later chunks - "pinned async obs copies into a GPU-resident rollout buffer" with the copy under the stepping
// instead of after it.  obs_out must be pinned host memory, obs_dev the device destination of env 0.
extern "C" int ppo_synth_env_step_upload(void *h, const int32_t *actions, uint8_t *obs_out, float *reward_out,
                                         uint8_t *done_out, int32_t *time_out, float *ep_score_out, int32_t *ep_len_out,
                                         void *obs_dev, int n_chunks, void *stream)
{
    auto *e = static_cast<SynthEnv *>(h);
    if (!e || !obs_out || !obs_dev) return ppo::fail(PPO_E_INVALID, "ppo_synth_env_step_upload: null");
    if (n_chunks < 1 || n_chunks > 16) return ppo::fail(PPO_E_INVALID, "ppo_synth_env_step_upload: 1..16 chunks");
    e->actions = actions;
    e->obs = obs_out;
    e->reward = reward_out;
    e->done = done_out;
    e->time_out = time_out;
    e->ep_score_out = ep_score_out;
    e->ep_len_out = ep_len_out;
    hipStream_t st = ppo::as_stream(stream);
    if (e->workers.empty()) {
        e->run_range(0, e->n_envs);
        hipError_t err = hipMemcpyAsync(obs_dev, obs_out, (size_t)e->n_envs * e->obs_bytes, hipMemcpyHostToDevice, st);
        return err == hipSuccess ? PPO_OK : ppo::fail(PPO_E_HIP, "ppo_synth_env_step_upload: %s", hipGetErrorString(err));
    }
    // blocks of 2 envs; a chunk is a whole number of blocks
    const int block = 2;
    const int n_blocks = (e->n_envs + block - 1) / block;
    const int bpc = (n_blocks + n_chunks - 1) / n_chunks;
    const int chunks = (n_blocks + bpc - 1) / bpc;
    e->block = block;
    e->blocks_per_chunk = bpc;
    e->n_chunks = chunks;
    e->next_block.store(0);
    for (int c = 0; c < chunks; ++c) {
        const int first = c * bpc, last = (c + 1) * bpc < n_blocks ? (c + 1) * bpc : n_blocks;
        e->chunk_left[c].store(last - first);
    }
    e->start();
    int rc = PPO_OK;
    for (int c = 0; c < chunks; ++c) {
        while (e->chunk_left[c].load(std::memory_order_acquire) > 0) {
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
        const int64_t lo = (int64_t)c * bpc * block;
        int64_t hi = (int64_t)(c + 1) * bpc * block;
        hi = hi < e->n_envs ? hi : e->n_envs;
        hipError_t err = hipMemcpyAsync(static_cast<uint8_t *>(obs_dev) + lo * e->obs_bytes, obs_out + lo * e->obs_bytes,
                                        (size_t)(hi - lo) * e->obs_bytes, hipMemcpyHostToDevice, st);
        if (err != hipSuccess && rc == PPO_OK) rc = ppo::fail(PPO_E_HIP, "ppo_synth_env_step_upload: %s", hipGetErrorString(err));
    }
    e->join();
    e->block = 0;
    return rc;
}

// Checkpointing (rl/hybridVecEnv.py:84-105 save_state / restore_state of the worker envs): the generator state of an
// env is its step count; time / score are the running episode's statistics.
extern "C" int ppo_synth_env_get_state(void *h, int64_t *steps_out, int32_t *time_out, float *score_out)
{
    auto *e = static_cast<SynthEnv *>(h);
    if (!e || !steps_out || !time_out || !score_out) return ppo::fail(PPO_E_INVALID, "ppo_synth_env_get_state: null");
    std::memcpy(steps_out, e->steps.data(), sizeof(int64_t) * e->n_envs);
    std::memcpy(time_out, e->time.data(), sizeof(int32_t) * e->n_envs);
    std::memcpy(score_out, e->score.data(), sizeof(float) * e->n_envs);
    return PPO_OK;
}

extern "C" int ppo_synth_env_set_state(void *h, const int64_t *steps, const int32_t *time, const float *score,
                                       uint8_t *obs_out)
{
    auto *e = static_cast<SynthEnv *>(h);
    if (!e || !steps || !time || !score || !obs_out) return ppo::fail(PPO_E_INVALID, "ppo_synth_env_set_state: null");
    for (int i = 0; i < e->n_envs; ++i)
        if (steps[i] < 0 || time[i] < 0) return ppo::fail(PPO_E_INVALID, "ppo_synth_env_set_state: negative counter");
    std::memcpy(e->steps.data(), steps, sizeof(int64_t) * e->n_envs);
    std::memcpy(e->time.data(), time, sizeof(int32_t) * e->n_envs);
    std::memcpy(e->score.data(), score, sizeof(float) * e->n_envs);
    e->obs = obs_out;  // the current observation is a function of (seed, env, steps): regenerate it
    for (int i = 0; i < e->n_envs; ++i) e->fill_obs(i);
    return PPO_OK;
}
