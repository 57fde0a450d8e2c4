// detect.hip -- K11: batched correct_yolo_boxes + do_nms_sort on the GPU, one workgroup per image.
//
// Reference behaviour reproduced (the reference runs this on the host for batch
// item 0 only):
//   correct_yolo_boxes   src/additionally.c:4281-4314   (double arithmetic kept)
//   do_nms_sort          src/box.c:296-328
//   nms_comparator       src/box.c:280-294
//   box_iou & friends    src/box.c:55-97
//
// do_nms_sort is a sequential algorithm over classes: for every class k it qsorts ALL
// detections by prob[k] descending (glibc's qsort is a stable merge sort, so ties keep the order
// the previous class left behind), then greedily zeroes prob[k] of every later box whose IoU with
// a surviving earlier box exceeds the threshold.  The final row order is whatever the last sort
// left.  To return *exactly* the reference's rows (values and order) the kernel carries the
// permutation through the classes the same way, but does only the work that can change it:
//   * a class in which no detection has prob > 0 leaves the order untouched (all keys equal,
//     stable sort) and suppresses nothing -> skipped via a per-image class bitmap;
//   * otherwise "stable sort by prob desc" == stable partition (prob > 0 first; probabilities are
//     never negative) + a stable rank sort of the m positive ones, m << count.
// The greedy pass works on 64 sorted pivots at a time: one wave settles the chunk internally with
// the dead-set as a 64-bit ballot mask (no barriers), then the whole workgroup applies the chunk's
// survivors to everything behind it.
//
// Input records come from compact_kernel (layers.hip) in atomic-slot order; r[5] carries the
// reference's scan-order key (head, cell, anchor), which the kernel sorts by first, so the result
// does not depend on the atomics.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace yl {

namespace {

constexpr int NMS_THREADS = 256;

struct BoxF { float x, y, w, h; };

__device__ __forceinline__ float overlap1d(float x1, float w1, float x2, float w2)
{
    const float l1 = __fsub_rn(x1, __fdiv_rn(w1, 2.f));
    const float l2 = __fsub_rn(x2, __fdiv_rn(w2, 2.f));
    const float left = l1 > l2 ? l1 : l2;
    const float r1 = __fadd_rn(x1, __fdiv_rn(w1, 2.f));
    const float r2 = __fadd_rn(x2, __fdiv_rn(w2, 2.f));
    const float right = r1 < r2 ? r1 : r2;
    return __fsub_rn(right, left);
}

// box_iou(a, b) = box_intersection / box_union, src/box.c:55-97 (float, no contraction)
__device__ __forceinline__ float box_iou_dev(const BoxF a, const BoxF b)
{
    const float w = overlap1d(a.x, a.w, b.x, b.w);
    const float h = overlap1d(a.y, a.h, b.y, b.h);
    const float inter = (w < 0 || h < 0) ? 0.f : __fmul_rn(w, h);
    const float uni = __fsub_rn(__fadd_rn(__fmul_rn(a.w, a.h), __fmul_rn(b.w, b.h)), inter);
    return __fdiv_rn(inter, uni);
}

// exclusive prefix sum of one int per thread over the block (NMS_THREADS), result via LDS
__device__ __forceinline__ int block_exclusive_scan(int v, int *scratch, int *total)
{
    const int t = threadIdx.x;
    scratch[t] = v;
    __syncthreads();
    for (int off = 1; off < NMS_THREADS; off <<= 1) {
        const int add = (t >= off) ? scratch[t - off] : 0;
        __syncthreads();
        scratch[t] += add;
        __syncthreads();
    }
    const int incl = scratch[t];
    *total = scratch[NMS_THREADS - 1];
    __syncthreads();
    return incl - v;
}

}  // namespace

// LDS layout (dynamic, 30 B per record slot), cap <= NMS_MAX_CAP:
//   float  bx[cap], by[cap], bw[cap], bh[cap]   corrected boxes, indexed by record slot
//   float  pk[cap]                              prob[k] by position (current class)
//   float  ps[cap]                              prob[k] of the sorted positive prefix
//   uint16 perm[2][cap]                         position -> record slot (double buffer)
//   uint16 sidx[cap]                            sorted positive prefix -> record slot
//   int    scan[NMS_THREADS], uint32 cmask[(classes+31)/32], misc
// MODE 0: everything in one workgroup per image (sequential over classes).
// MODE 2 / nms_class_kernel / MODE 1: the same result with the suppression of every (image, class)
//   pair in its own workgroup -- MODE 2 prepares (corrected boxes back into the rows, position in the
//   reference's initial order into column 5, class bitmap and `total` into `meta`), nms_class_kernel
//   suppresses (marks by flipping the sign of the probability, the magnitude stays readable for the
//   other classes' tie-breaks), MODE 1 replays only the ORDER (partition + rank sort per class on the
//   original magnitudes) and emits.
template <int MODE>
__global__ __launch_bounds__(NMS_THREADS) void nms_kernel(float *__restrict__ rec, const int *__restrict__ counts,
                                                          int cap, int classes, int row_stride, float nms,
                                                          int netw, int neth, ImgDims dims,
                                                          int relative, int letter,
                                                          float *__restrict__ rec_out, int *__restrict__ counts_out,
                                                          unsigned *__restrict__ meta)
{
    extern __shared__ unsigned char smem[];
    float *bx = (float *)smem;
    float *by = bx + cap;
    float *bw = by + cap;
    float *bh = bw + cap;
    float *pk = bh + cap;
    float *ps = pk + cap;
    uint16_t *permA = (uint16_t *)(ps + cap);
    uint16_t *permB = permA + cap;
    uint16_t *sidx = permB + cap;
    int *scan = (int *)(sidx + cap + (cap & 1));
    uint32_t *cmask = (uint32_t *)(scan + NMS_THREADS);
    const int cwords = (classes + 31) >> 5;
    int *misc = (int *)(cmask + cwords);

    const int b = blockIdx.x;
    const int t = threadIdx.x;
    const int raw = counts[b];
    const int cnt = raw < cap ? raw : cap;
    float *in = rec + (size_t)b * cap * row_stride;          // scratch rows: prob columns are zeroed in place
    float *out = rec_out + (size_t)b * cap * row_stride;
    if (t == 0 && counts_out) counts_out[b] = raw;

    // ---- correct_yolo_boxes parameters (block-uniform), src/additionally.c:4281-4314 ----
    const uint32_t packed = dims.mode == 2 ? dims.wh[b] : dims.wh[0];
    const int iw = dims.mode ? (int)(packed & 0xFFFFu) : netw;
    const int ih = dims.mode ? (int)(packed >> 16) : neth;
    int new_w, new_h;
    if (letter) {
        if (__fdiv_rn((float)netw, (float)iw) < __fdiv_rn((float)neth, (float)ih)) { new_w = netw; new_h = (ih * netw) / iw; }
        else { new_h = neth; new_w = (iw * neth) / ih; }
    } else { new_w = netw; new_h = neth; }
    const double off_x = (double)(netw - new_w) / 2. / (double)netw;
    const double off_y = (double)(neth - new_h) / 2. / (double)neth;
    const double sc_x = (double)__fdiv_rn((float)new_w, (float)netw);
    const double sc_y = (double)__fdiv_rn((float)new_h, (float)neth);
    const float mul_w = __fdiv_rn((float)netw, (float)new_w);
    const float mul_h = __fdiv_rn((float)neth, (float)new_h);

    for (int i = t; i < cwords; i += NMS_THREADS) cmask[i] = 0;
    if (t == 0) misc[0] = 0;
    __syncthreads();

    // ---- load: corrected boxes, bitmap of classes with any prob > 0, zero-objectness flag ----
    for (int s = t; s < cnt; s += NMS_THREADS) {
        const float *r = in + (size_t)s * row_stride;
        float x = r[0], y = r[1], w = r[2], h = r[3];
        if (MODE != 1) {
            x = (float)(((double)r[0] - off_x) / sc_x);
            y = (float)(((double)r[1] - off_y) / sc_y);
            w = __fmul_rn(r[2], mul_w);
            h = __fmul_rn(r[3], mul_h);
        }
        if (MODE != 1 && !relative) {
            x = __fmul_rn(x, (float)iw); w = __fmul_rn(w, (float)iw);
            y = __fmul_rn(y, (float)ih); h = __fmul_rn(h, (float)ih);
        }
        bx[s] = x; by[s] = y; bw[s] = w; bh[s] = h;
        if (MODE != 1 && r[4] == 0.f) misc[0] = 1;
        pk[s] = r[5];                                          // scan-order key (MODE 1: initial position)
        for (int j = 0; j < classes; ++j)
            if (r[6 + j] != 0.f) atomicOr(&cmask[j >> 5], 1u << (j & 31));     // suppressed (negated) ones count too
    }
    __syncthreads();

    // ---- initial order = the reference's scan order; rank sort, keys are unique ----
    for (int s = t; s < cnt; s += NMS_THREADS) {
        const float key = pk[s];
        int rank = 0;
        for (int j = 0; j < cnt; ++j) rank += (pk[j] < key) ? 1 : 0;
        permA[rank] = (uint16_t)s;
    }
    __syncthreads();

    int total = cnt;
    uint16_t *perm = permA, *perm_nxt = permB;
    if (nms > 0) {
        // "move zero-objectness detections to the end" (src/box.c:300-309): sequential swaps.  Only
        // reachable with thresh < 0 and a logistic that underflowed to 0; kept for exactness.
        if (MODE != 1 && misc[0]) {
            if (t == 0) {
                int k = cnt - 1;
                for (int i = 0; i <= k; ++i) {
                    if (in[(size_t)perm[i] * row_stride + 4] == 0.f) {
                        const uint16_t sw = perm[i]; perm[i] = perm[k]; perm[k] = sw;
                        --k; --i;
                    }
                }
                misc[1] = k + 1;
            }
            __syncthreads();
            total = misc[1];
            for (int p = total + t; p < cnt; p += NMS_THREADS) perm_nxt[p] = perm[p];   // the tail never moves again
            __syncthreads();
        }

        if (MODE == 1) {
            total = (int)meta[(size_t)b * (1 + cwords)];
            for (int p = total + t; p < cnt; p += NMS_THREADS) perm_nxt[p] = perm[p];
            __syncthreads();
        }
        if (MODE == 2) {
            for (int s = t; s < cnt; s += NMS_THREADS) {
                float *r = in + (size_t)s * row_stride;
                r[0] = bx[s]; r[1] = by[s]; r[2] = bw[s]; r[3] = bh[s];
            }
            for (int p = t; p < cnt; p += NMS_THREADS) in[(size_t)perm[p] * row_stride + 5] = (float)p;
            unsigned *mb = meta + (size_t)b * (1 + cwords);
            if (t == 0) mb[0] = (unsigned)total;
            for (int i = t; i < cwords; i += NMS_THREADS) mb[1 + i] = cmask[i];
            return;
        }

        for (int k = 0; k < classes; ++k) {
            if (!((cmask[k >> 5] >> (k & 31)) & 1u)) continue;          // block-uniform
            // thread t owns positions [p0,p1) so the partition below is stable
            const int per = (total + NMS_THREADS - 1) / NMS_THREADS;
            const int p0 = t * per;
            const int p1 = (p0 + per < total) ? p0 + per : total;
            int mine = 0;
            for (int p = p0; p < p1; ++p) {
                float v = in[(size_t)perm[p] * row_stride + 6 + k];
                if (MODE == 1) v = fabsf(v);                 // the order is defined on the original values
                pk[p] = v;
                mine += (v > 0.f) ? 1 : 0;
            }
            int m = 0;
            const int before = block_exclusive_scan(mine, scan, &m);
            {
                int pos_i = before, neg_i = m + (p0 - before);
                for (int p = p0; p < p1; ++p) {
                    if (pk[p] > 0.f) { sidx[pos_i] = perm[p]; ps[pos_i] = pk[p]; ++pos_i; }
                    else perm_nxt[neg_i++] = perm[p];
                }
            }
            __syncthreads();
            // stable rank sort of the positive prefix by prob desc -> perm_nxt[0,m), pk[0,m)
            for (int i = t; i < m; i += NMS_THREADS) {
                const float v = ps[i];
                int rank = 0;
                for (int j = 0; j < m; ++j) {
                    const float u = ps[j];
                    rank += (u > v || (u == v && j < i)) ? 1 : 0;
                }
                perm_nxt[rank] = sidx[i];
                pk[rank] = v;
            }
            __syncthreads();
            // greedy suppression over the sorted prefix (positions >= m have prob 0: `continue`),
            // 64 pivots at a time: wave 0 settles the chunk internally with the dead-set as a
            // ballot mask (no barriers), then every thread applies the chunk's survivors to the
            // elements behind it -- one barrier pair per 64 pivots instead of one per pivot.
            for (int c0 = 0; MODE == 0 && c0 < m; c0 += 64) {
                const int cn = (m - c0 < 64) ? m - c0 : 64;
                if (t < 64) {
                    const bool valid = t < cn;
                    const int slot = valid ? perm_nxt[c0 + t] : 0;
                    const BoxF me = { bx[slot], by[slot], bw[slot], bh[slot] };
                    // bit j set: element c0+j is dead (suppressed by an earlier chunk, or below)
                    unsigned long long dead = __ballot(valid && pk[c0 + t] == 0.f);
                    for (int i = 0; i < cn; ++i) {
                        if ((dead >> i) & 1ull) continue;
                        BoxF a;
                        a.x = __shfl(me.x, i); a.y = __shfl(me.y, i); a.w = __shfl(me.w, i); a.h = __shfl(me.h, i);
                        const bool kill = valid && t > i && (box_iou_dev(a, me) > nms);
                        dead |= __ballot(kill);
                    }
                    if (valid && ((dead >> t) & 1ull) && pk[c0 + t] != 0.f) {
                        pk[c0 + t] = 0.f;
                        in[(size_t)slot * row_stride + 6 + k] = 0.f;
                    }
                    if (t == 0) { misc[2] = (int)(unsigned)(dead & 0xFFFFFFFFull); misc[3] = (int)(unsigned)(dead >> 32); }
                }
                if (c0 + 64 >= m) break;              // block-uniform: nothing behind this chunk
                __syncthreads();
                const unsigned long long dead = ((unsigned long long)(unsigned)misc[3] << 32) | (unsigned)misc[2];
                for (int j = c0 + 64 + t; j < m; j += NMS_THREADS) {
                    if (pk[j] == 0.f) continue;
                    const int sj = perm_nxt[j];
                    const BoxF me = { bx[sj], by[sj], bw[sj], bh[sj] };
                    bool kill = false;
                    for (int i = 0; i < 64 && !kill; ++i) {
                        if ((dead >> i) & 1ull) continue;
                        const int si = perm_nxt[c0 + i];
                        const BoxF a = { bx[si], by[si], bw[si], bh[si] };
                        kill = box_iou_dev(a, me) > nms;
                    }
                    if (kill) { pk[j] = 0.f; in[(size_t)sj * row_stride + 6 + k] = 0.f; }
                }
                __syncthreads();
            }
            __syncthreads();
            uint16_t *sw = perm; perm = perm_nxt; perm_nxt = sw;
        }
    }
    __threadfence_block();
    __syncthreads();

    // ---- emit rows in the final order (coalesced over the flattened [row][column] index) ----
    const float sort_class = (nms > 0) ? (float)(classes - 1) : 0.f;
    const int n_out = cnt * row_stride;
    for (int e = t; e < n_out; e += NMS_THREADS) {
        const int p = e / row_stride, c = e - p * row_stride;
        const int s = perm[p];
        float v;
        if (c == 0) v = bx[s];
        else if (c == 1) v = by[s];
        else if (c == 2) v = bw[s];
        else if (c == 3) v = bh[s];
        else if (c == 5) v = (p < total) ? sort_class : 0.f;     // the unsorted tail keeps calloc's 0
        else v = in[(size_t)s * row_stride + c];
        if (MODE == 1 && c >= 6 && v < 0.f) v = 0.f;               // marked by nms_class_kernel
        out[e] = v;
    }
}

// Suppression of ONE class of ONE image (grid = classes x batch): the positives of class k are sorted
// the way the reference's k-th qsort leaves them -- by probability, ties by the order the earlier
// classes produced, which is a pure function of the ORIGINAL probabilities: a before b iff at the
// highest class c < k where |p_c| differs a's is larger, else by the initial position -- and the
// greedy pass runs on that list.  A suppressed probability gets its sign flipped.
__global__ __launch_bounds__(NMS_THREADS) void nms_class_kernel(float *__restrict__ rec, const int *__restrict__ counts,
                                                                int cap, int classes, int row_stride, float nms,
                                                                const unsigned *__restrict__ meta)
{
    extern __shared__ unsigned char smem[];
    float *gp = (float *)smem;                 // gathered probabilities
    float *gpos = gp + cap;                    // their initial positions
    float *sp = gpos + cap;                    // sorted probabilities (sign flipped = suppressed)
    float *pbox = sp + cap;                    // 64 pivot boxes
    uint16_t *gslot = (uint16_t *)(pbox + 256);
    uint16_t *sidx = gslot + cap;
    int *misc = (int *)(sidx + cap + ((2 * cap) & 1));

    const int k = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
    const int cwords = (classes + 31) >> 5;
    const unsigned *mb = meta + (size_t)b * (1 + cwords);
    if (!((mb[1 + (k >> 5)] >> (k & 31)) & 1u)) return;
    const int total = (int)mb[0];
    const int raw = counts[b];
    const int cnt = raw < cap ? raw : cap;
    float *in = rec + (size_t)b * cap * row_stride;

    if (t == 0) misc[0] = 0;
    __syncthreads();
    for (int s = t; s < cnt; s += NMS_THREADS) {
        const float p = in[(size_t)s * row_stride + 6 + k];
        const float pos = in[(size_t)s * row_stride + 5];
        if (p > 0.f && pos < (float)total) {
            const int i = atomicAdd(&misc[0], 1);
            gp[i] = p; gpos[i] = pos; gslot[i] = (uint16_t)s;
        }
    }
    __syncthreads();
    const int m = misc[0];
    if (m <= 1) return;
    for (int i = t; i < m; i += NMS_THREADS) {
        const float v = gp[i];
        const size_t ri = (size_t)gslot[i] * row_stride;
        int rank = 0;
        for (int j = 0; j < m; ++j) {
            const float u = gp[j];
            if (u > v) { ++rank; continue; }
            if (u != v || j == i) continue;
            // tie: the order class k-1's sort left behind
            const size_t rj = (size_t)gslot[j] * row_stride;
            bool before = gpos[j] < gpos[i];
            for (int c = k - 1; c >= 0; --c) {
                const float a = fabsf(in[rj + 6 + c]), d = fabsf(in[ri + 6 + c]);
                if (a != d) { before = a > d; break; }
            }
            rank += before ? 1 : 0;
        }
        sp[rank] = v;
        sidx[rank] = gslot[i];
    }
    __syncthreads();
    for (int c0 = 0; c0 < m; c0 += 64) {
        const int cn = (m - c0 < 64) ? m - c0 : 64;
        if (t < 64) {
            const bool valid = t < cn;
            const int slot = valid ? sidx[c0 + t] : 0;
            const float *r = in + (size_t)slot * row_stride;
            const BoxF me = { r[0], r[1], r[2], r[3] };
            unsigned long long dead = __ballot(valid && sp[c0 + t] < 0.f);
            for (int i = 0; i < cn; ++i) {
                if ((dead >> i) & 1ull) continue;
                BoxF a;
                a.x = __shfl(me.x, i); a.y = __shfl(me.y, i); a.w = __shfl(me.w, i); a.h = __shfl(me.h, i);
                const bool kill = valid && t > i && (box_iou_dev(a, me) > nms);
                dead |= __ballot(kill);
            }
            if (valid && ((dead >> t) & 1ull) && sp[c0 + t] > 0.f) {
                sp[c0 + t] = -sp[c0 + t];
                in[(size_t)slot * row_stride + 6 + k] = sp[c0 + t];
            }
            if (valid) { pbox[t * 4 + 0] = me.x; pbox[t * 4 + 1] = me.y; pbox[t * 4 + 2] = me.w; pbox[t * 4 + 3] = me.h; }
            if (t == 0) { misc[2] = (int)(unsigned)(dead & 0xFFFFFFFFull); misc[3] = (int)(unsigned)(dead >> 32); }
        }
        if (c0 + 64 >= m) break;
        __syncthreads();
        const unsigned long long dead = ((unsigned long long)(unsigned)misc[3] << 32) | (unsigned)misc[2];
        for (int j = c0 + 64 + t; j < m; j += NMS_THREADS) {
            if (sp[j] < 0.f) continue;
            const int sj = sidx[j];
            const float *r = in + (size_t)sj * row_stride;
            const BoxF me = { r[0], r[1], r[2], r[3] };
            bool kill = false;
            for (int i = 0; i < 64 && !kill; ++i) {
                if ((dead >> i) & 1ull) continue;
                const BoxF a = { pbox[i * 4 + 0], pbox[i * 4 + 1], pbox[i * 4 + 2], pbox[i * 4 + 3] };
                kill = box_iou_dev(a, me) > nms;
            }
            if (kill) { sp[j] = -sp[j]; in[(size_t)sj * row_stride + 6 + k] = sp[j]; }
        }
        __syncthreads();
    }
}

static size_t nms_class_lds_bytes(int cap)
{
    return (size_t)(3 * cap + 256) * sizeof(float) + (size_t)(2 * cap + 2) * sizeof(uint16_t) + 4 * sizeof(int);
}

size_t nms_lds_bytes(int cap, int classes)
{
    size_t n = (size_t)cap * 6 * sizeof(float) + (size_t)(3 * cap + (cap & 1)) * sizeof(uint16_t);
    n += NMS_THREADS * sizeof(int) + (size_t)((classes + 31) / 32) * sizeof(uint32_t) + 4 * sizeof(int);
    return n;
}


int launch_nms(float *rec_scratch, const int *counts, int B, int cap, int classes, float nms, int netw, int neth,
               const ImgDims &dims, int relative, int letter, float *rec_out, int *counts_out, unsigned *meta,
               int mode, void *stream)
{
    if (cap > NMS_MAX_CAP) return (int)hipErrorInvalidValue;
    const int row_stride = 6 + classes;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = nms_lds_bytes(cap, classes);
    const size_t lds_c = nms_class_lds_bytes(cap);
    if (lds > 64 * 1024 || lds_c > 64 * 1024) {
        // beyond the default 64 KB of dynamic LDS per workgroup (gfx950 has 160 KB per CU); per device, cheap
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&nms_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&nms_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&nms_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&nms_class_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c);
        if (e != hipSuccess) return (int)e;
    }
    if (mode == 0 || !(nms > 0) || !meta) {   // mode 1 = (image, class)-parallel suppression
        hipLaunchKernelGGL(nms_kernel<0>, dim3(B), dim3(NMS_THREADS), lds, s, rec_scratch, counts, cap, classes,
                           row_stride, nms, netw, neth, dims, relative, letter, rec_out, counts_out, meta);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(nms_kernel<2>, dim3(B), dim3(NMS_THREADS), lds, s, rec_scratch, counts, cap, classes,
                       row_stride, nms, netw, neth, dims, relative, letter, rec_out, counts_out, meta);
    hipLaunchKernelGGL(nms_class_kernel, dim3(classes, B), dim3(NMS_THREADS), lds_c, s,
                       rec_scratch, counts, cap, classes, row_stride, nms, meta);
    hipLaunchKernelGGL(nms_kernel<1>, dim3(B), dim3(NMS_THREADS), lds, s, rec_scratch, counts, cap, classes,
                       row_stride, nms, netw, neth, dims, relative, letter, rec_out, counts_out, meta);
    return (int)hipGetLastError();
}

}  // namespace yl
