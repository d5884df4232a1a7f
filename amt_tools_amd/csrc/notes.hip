// Note decoding on the device: binary piano rolls -> (onset frame, offset frame) events per key.
// Replaces the per-event Python `while` loop of tools.multi_pitch_to_notes
// (amt_tools/tools/utils.py:369-471, with multi_pitch_to_onsets :2381-2412) that
// transcribe.NoteTranscriber (amt_tools/transcribe.py:420-481,722-763) runs on every track:
//     mp = mp OR onsets;  impulses = positive first difference of onsets (first frame counts);
//     every impulse (key, t) walks forward until the end, an inactive frame or the next impulse.
// Integer/boolean work, bit-exact by construction.  One wave64 per (clip, key) row scans the row backwards
// in 64-frame chunks, carrying the index of the next "stop" frame; impulses are compacted with a ballot
// prefix count.  HBM-bound: reads 2 x 4 B per cell once, writes 8 B per note.

#include "amtx_kernels.h"

namespace {

__global__ __launch_bounds__(256) void notes_kernel(const float* __restrict__ onsets, const float* __restrict__ mp, int rows, int T,
                                                    int use_onsets, int cap, int2* __restrict__ pairs, int* __restrict__ counts) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* on = (use_onsets ? onsets : mp) + (int64_t)row * T;
    const float* act = mp + (int64_t)row * T;
    int2* out = pairs + (int64_t)row * cap;
    int next_stop = T;          // first stop frame strictly after the current chunk
    int n = 0;                  // events written so far (descending frame order)
    const int nchunks = (T + 63) >> 6;
    for (int ch = nchunks - 1; ch >= 0; --ch) {
        const int t = ch * 64 + lane;
        const bool valid = t < T;
        const float o = valid ? on[t] : 0.f;
        const float oprev = (valid && t > 0) ? on[t - 1] : 0.f;
        const bool imp = valid && ((t == 0) ? (o > 0.f) : (o - oprev > 0.f));
        // onsets given: activity = onsets OR multi_pitch; derived onsets: activity = multi_pitch (utils.py:403-411)
        const bool active = valid && ((act[t] != 0.f) || (use_onsets && o != 0.f));
        const bool stop = valid && (!active || imp);
        const unsigned long long smask = __ballot(stop);
        const unsigned long long above = (lane == 63) ? 0ull : (smask >> (lane + 1));
        const int my_next = above ? (t + 1 + __builtin_ctzll(above)) : next_stop;
        const unsigned long long imask = __ballot(imp);
        if (imp) {
            // events of this chunk in descending frame order after those of later chunks
            const int higher = __builtin_popcountll(lane == 63 ? 0ull : (imask >> (lane + 1)));
            const int slot = n + higher;
            if (slot < cap) out[slot] = make_int2(t, my_next);
        }
        n += __builtin_popcountll(imask);
        if (smask) next_stop = ch * 64 + __builtin_ctzll(smask);
    }
    if (lane == 0) counts[row] = n;
}

}  // namespace

extern "C" int amtx_notes_decode(const float* onsets, const float* multi_pitch, int batch, int keys, int num_frames, int capacity,
                                 int32_t* pairs, int32_t* counts, void* stream) {
    AMTX_REQUIRE(multi_pitch && pairs && counts, "amtx_notes_decode: null pointer");
    AMTX_REQUIRE(batch > 0 && keys > 0 && num_frames > 0 && capacity > 0, "amtx_notes_decode: bad sizes");
    const int rows = batch * keys;
    hipLaunchKernelGGL(notes_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, onsets, multi_pitch, rows,
                       num_frames, onsets ? 1 : 0, capacity, reinterpret_cast<int2*>(pairs), counts);
    AMTX_CHECK_LAUNCH();
    return AMTX_OK;
}
