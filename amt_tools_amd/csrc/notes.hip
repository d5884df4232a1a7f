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

// ---------------------------------------------------------------------------------------------------------------------------
// Note rows on the device.  amtx_notes_decode leaves, per (clip, key), its events in descending frame order in a padded array.
// The three kernels below turn them into what transcribe.py's host code used to assemble clip by clip: ONE dense array of note rows
// [onset_s, offset_s, midi_pitch] (float64, the reference's batched-notes layout, tools/utils.py:135-165) in np.nonzero order -- key
// ascending, onset frame ascending, exactly the order in which tools.multi_pitch_to_notes (utils.py:369-471) appends its notes --
// with frames converted to seconds through the clip's own (extended) time grid, a contiguous copy of the onset column (the key of the
// reference's three argsorts) and a per-clip offset table.  What stays on the host is the reference's row ORDER among equal onsets:
// it is whatever NumPy's unstable argsort makes of it, three times over (sort_notes, utils.py:2713-2746), and only NumPy can say.

// clip_counts[b] = number of events of clip b (one wave per clip)
__global__ __launch_bounds__(256) void notes_clip_count_kernel(const int* __restrict__ counts, int batch, int keys, int cap, int* __restrict__ clip_counts) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= batch) return;
    int n = 0;
    for (int k = lane; k < keys; k += 64) n += min(counts[(int64_t)b * keys + k], cap);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
    if (lane == 0) clip_counts[b] = n;
}

// exclusive scan of clip_counts[0 .. batch) into offsets[0 .. batch] (one block; 64-bit running sum, offsets clamp at INT32_MAX)
__global__ __launch_bounds__(1024) void notes_scan_kernel(const int* __restrict__ clip_counts, int batch, int* __restrict__ offsets) {
    __shared__ long long part[1024];
    __shared__ long long carry;
    const int tid = threadIdx.x;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < batch; base += 1024) {
        const int i = base + tid;
        const long long v = i < batch ? clip_counts[i] : 0;
        part[tid] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const long long add = tid >= o ? part[tid - o] : 0;
            __syncthreads();
            part[tid] += add;
            __syncthreads();
        }
        const long long excl = carry + part[tid] - v;
        if (i < batch) offsets[i] = (int)min(excl, (long long)0x7fffffff);
        __syncthreads();
        if (tid == 1023) carry += part[1023];
        __syncthreads();
    }
    if (tid == 0) offsets[batch] = (int)min(carry, (long long)0x7fffffff);
}

// one wave per (clip, key): its events, ascending frame, behind those of the clip's lower keys
__global__ __launch_bounds__(256) void notes_rows_kernel(const int2* __restrict__ pairs, const int* __restrict__ counts, int batch, int keys, int cap,
                                                         const double* __restrict__ times_ext, int64_t times_stride, int low,
                                                         const int* __restrict__ offsets, double* __restrict__ rows, double* __restrict__ onset_col,
                                                         int64_t rows_capacity) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= batch * keys) return;
    const int b = row / keys, key = row - b * keys;
    int before = 0;
    for (int k = lane; k < key; k += 64) before += min(counts[(int64_t)b * keys + k], cap);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o);
    const int n = min(counts[row], cap);
    const int64_t base = (int64_t)offsets[b] + before;
    const double* tg = times_ext + (int64_t)b * times_stride;
    const int2* ev = pairs + (int64_t)row * cap;
    for (int j = lane; j < n; j += 64) {
        const int64_t dst = base + j;
        if (dst >= rows_capacity) break;
        const int2 e = ev[n - 1 - j];                 // the decoder wrote descending frames
        const double on = tg[e.x];
        rows[dst * 3 + 0] = on;
        rows[dst * 3 + 1] = tg[e.y];
        rows[dst * 3 + 2] = (double)(key + low);
        onset_col[dst] = on;
    }
}

}  // namespace

extern "C" int amtx_notes_rows(const int32_t* pairs, const int32_t* counts, int batch, int keys, int capacity, const double* times_ext,
                               int64_t times_stride, int low_pitch, double* rows, double* onset_col, int64_t rows_capacity, int32_t* clip_offsets,
                               void* stream_) {
    AMTX_REQUIRE(pairs && counts && times_ext && rows && onset_col && clip_offsets, "amtx_notes_rows: null pointer");
    AMTX_REQUIRE(batch > 0 && keys > 0 && capacity > 0 && rows_capacity > 0 && times_stride >= 0, "amtx_notes_rows: bad sizes");
    AMTX_REQUIRE((int64_t)batch * keys < (1ll << 31), "amtx_notes_rows: too many rows");
    hipStream_t stream = (hipStream_t)stream_;
    // clip_offsets[1 .. batch] doubles as the per-clip count buffer: the scan kernel loads a 1024-clip chunk of counts into LDS, and only
    // then writes that chunk's offsets, which sit one element LOWER (offset i overwrites count i - 1: this chunk's, already in LDS, or
    // the previous chunk's last one); the total lands in clip_offsets[batch] = count slot batch - 1, read long before.
    int* cc = clip_offsets + 1;
    hipLaunchKernelGGL(notes_clip_count_kernel, dim3((unsigned)((batch + 3) / 4)), dim3(256), 0, stream, counts, batch, keys, capacity, cc);
    AMTX_CHECK_LAUNCH();
    hipLaunchKernelGGL(notes_scan_kernel, dim3(1), dim3(1024), 0, stream, (const int*)cc, batch, clip_offsets);
    AMTX_CHECK_LAUNCH();
    const int nrows = batch * keys;
    hipLaunchKernelGGL(notes_rows_kernel, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, stream, reinterpret_cast<const int2*>(pairs), counts, batch, keys,
                       capacity, times_ext, times_stride, low_pitch, (const int*)clip_offsets, rows, onset_col, rows_capacity);
    AMTX_CHECK_LAUNCH();
    return AMTX_OK;
}

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
