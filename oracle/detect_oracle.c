/*
 * oracle/detect_oracle.c -- CPU RESTATEMENT of the reference's detection decode + NMS.
 *
 * TEST INFRASTRUCTURE ONLY (part of liboracle.so).  Nothing in the product may include, link
 * or call this file; tests/ use it as the checker of the GPU decode/NMS kernels
 * (yolo2_light_amd/csrc/layers.hip K10, detect.hip K11).
 *
 * Restates, with the reference's types and evaluation order so that rows come out bit-identical
 * (order included) on the same head tensors:
 *   get_network_boxes / make_network_boxes / num_detections   src/additionally.c:4403 / 4238 / 4222
 *   yolo_num_detections / get_yolo_detections / get_yolo_box   src/additionally.c:4207 / 4328 / 4317
 *   correct_yolo_boxes                                         src/additionally.c:4281
 *   custom_get_region_detections                               src/additionally.c:4363
 *   get_region_boxes_cpu / get_region_box_cpu                  src/yolov2_forward_network.c:664 / 653
 *   do_nms_sort / nms_comparator / box_iou                     src/box.c:296 / 280 / 94
 * Pinned row for row against the reference itself (oracle/_ref/libyolo2ref.so, ref_get_detections)
 * by tests/test_detect_host.py.  Extension: `image` selects the batch item (the reference is
 * hard-wired to item 0, src/additionally.c:4213,4338).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define HEAD_REGION 21
#define HEAD_YOLO 22

/* one detection head, fields as in the reference's `layer` */
typedef struct oracle_head {
    int type;               /* HEAD_YOLO / HEAD_REGION (== LAYER_TYPE values) */
    int w, h, n, classes, outputs;
    const float *output;    /* l.output: [batch][outputs] */
    const int *mask;        /* yolo: l.mask[n] */
    const float *anchors;   /* l.biases */
    const int *tree_parent; /* region with l.softmax_tree: t->parent[classes], else NULL */
} oracle_head;

typedef struct { float x, y, w, h; } obox;

/* field order and size of the reference's `detection` (src/box.h:9-17): qsort moves whole
 * elements, so the geometry must match for the merge to visit them identically */
typedef struct {
    obox bbox;
    int classes;
    float *prob;
    float *mask;
    float objectness;
    int sort_class;
} odet;

static int entry_index(const oracle_head *l, int batch, int location, int entry)
{
    int n = location / (l->w * l->h);
    int loc = location % (l->w * l->h);
    return batch * l->outputs + n * l->w * l->h * (4 + l->classes + 1) + entry * l->w * l->h + loc;
}

static float o_overlap(float x1, float w1, float x2, float w2)
{
    float l1 = x1 - w1 / 2;
    float l2 = x2 - w2 / 2;
    float left = l1 > l2 ? l1 : l2;
    float r1 = x1 + w1 / 2;
    float r2 = x2 + w2 / 2;
    float right = r1 < r2 ? r1 : r2;
    return right - left;
}

static float o_iou(obox a, obox b)
{
    float w = o_overlap(a.x, a.w, b.x, b.w);
    float h = o_overlap(a.y, a.h, b.y, b.h);
    float inter = (w < 0 || h < 0) ? 0 : w * h;
    float uni = a.w * a.h + b.w * b.h - inter;
    return inter / uni;
}

static int o_cmp(const void *pa, const void *pb)
{
    const odet *a = (const odet *)pa, *b = (const odet *)pb;
    float diff = (b->sort_class >= 0) ? a->prob[b->sort_class] - b->prob[b->sort_class]
                                       : a->objectness - b->objectness;
    if (diff < 0) return 1;
    if (diff > 0) return -1;
    return 0;
}

static void o_correct(odet *dets, int n, int w, int h, int netw, int neth, int relative, int letter)
{
    int new_w = netw, new_h = neth, i;
    if (letter) {
        if (((float)netw / w) < ((float)neth / h)) { new_w = netw; new_h = (h * netw) / w; }
        else { new_h = neth; new_w = (w * neth) / h; }
    }
    for (i = 0; i < n; ++i) {
        obox b = dets[i].bbox;
        b.x = (b.x - (netw - new_w) / 2. / netw) / ((float)new_w / netw);
        b.y = (b.y - (neth - new_h) / 2. / neth) / ((float)new_h / neth);
        b.w *= (float)netw / new_w;
        b.h *= (float)neth / new_h;
        if (!relative) { b.x *= w; b.w *= w; b.y *= h; b.h *= h; }
        dets[i].bbox = b;
    }
}

/* rows[max_rows][6+classes] = x y w h objectness sort_class prob[classes]; returns the number of
 * detections the reference returns (may exceed max_rows), < 0 on a bad argument. */
int oracle_get_boxes(const oracle_head *heads, int n_heads, int netw, int neth, int image, int w, int h,
                     float thresh, int relative, int letter, float nms, float *rows, int max_rows)
{
    int hi, i, j, k, n, nboxes = 0, total, classes;
    odet *dets, *cur;
    float *probs;
    if (!heads || n_heads <= 0 || image < 0) return -1;
    classes = heads[n_heads - 1].classes;

    for (hi = 0; hi < n_heads; ++hi) {                      /* num_detections */
        const oracle_head *l = &heads[hi];
        if (l->type == HEAD_YOLO) {
            for (i = 0; i < l->w * l->h; ++i)
                for (n = 0; n < l->n; ++n)
                    if (l->output[entry_index(l, image, n * l->w * l->h + i, 4)] > thresh) ++nboxes;
        } else if (l->type == HEAD_REGION) {
            nboxes += l->w * l->h * l->n;
        } else return -1;
    }
    dets = (odet *)calloc(nboxes > 0 ? nboxes : 1, sizeof(odet));
    probs = (float *)calloc((size_t)(nboxes > 0 ? nboxes : 1) * (classes > 0 ? classes : 1), sizeof(float));
    for (i = 0; i < nboxes; ++i) dets[i].prob = probs + (size_t)i * classes;

    cur = dets;
    for (hi = 0; hi < n_heads; ++hi) {
        const oracle_head *l = &heads[hi];
        if (l->type == HEAD_YOLO) {                         /* get_yolo_detections */
            const float *p = l->output;
            const int lwh = l->w * l->h;
            int count = 0;
            for (i = 0; i < lwh; ++i) {
                int row = i / l->w, col = i % l->w;
                for (n = 0; n < l->n; ++n) {
                    int obj_index = entry_index(l, image, n * lwh + i, 4);
                    float objectness = p[obj_index];
                    if (objectness > thresh) {
                        int box_index = entry_index(l, image, n * lwh + i, 0);
                        int an = l->mask[n];
                        obox b;
                        b.x = (col + p[box_index + 0 * lwh]) / l->w;
                        b.y = (row + p[box_index + 1 * lwh]) / l->h;
                        b.w = exp(p[box_index + 2 * lwh]) * l->anchors[2 * an] / netw;
                        b.h = exp(p[box_index + 3 * lwh]) * l->anchors[2 * an + 1] / neth;
                        cur[count].bbox = b;
                        cur[count].objectness = objectness;
                        cur[count].classes = l->classes;
                        for (j = 0; j < l->classes && j < classes; ++j) {
                            float prob = objectness * p[entry_index(l, image, n * lwh + i, 4 + 1 + j)];
                            cur[count].prob[j] = (prob > thresh) ? prob : 0;
                        }
                        ++count;
                    }
                }
            }
            o_correct(cur, count, w, h, netw, neth, relative, letter);
            cur += count;
        } else {                                            /* custom_get_region_detections */
            const float *p = l->output + (size_t)image * l->outputs;
            const int tot = l->w * l->h * l->n;
            for (i = 0; i < l->w * l->h; ++i) {
                int row = i / l->w, col = i % l->w;
                for (n = 0; n < l->n; ++n) {
                    int index = i * l->n + n;
                    int box_index = index * (l->classes + 5);
                    float scale = p[box_index + 4];
                    obox b;
                    /* logistic_activate: 1./(1. + exp(-x)) in double, returned as float */
                    float lx = 1. / (1. + exp(-p[box_index + 0]));
                    float ly = 1. / (1. + exp(-p[box_index + 1]));
                    b.x = (col + lx) / l->w;
                    b.y = (row + ly) / l->h;
                    b.w = expf(p[box_index + 2]) * l->anchors[2 * n] / l->w;
                    b.h = expf(p[box_index + 3]) * l->anchors[2 * n + 1] / l->h;
                    cur[index].classes = l->classes;
                    cur[index].bbox = b;                     /* get_region_boxes_cpu runs with w = h = 1 */
                    cur[index].objectness = 1;
                    if (l->tree_parent) {
                        /* Yolo 9000, src/yolov2_forward_network.c:690-712: hierarchy_predictions
                         * (src/additionally.c:1878, only_leaves = 0) on a copy -- the reference multiplies
                         * l.output in place -- then from the last class down the first one above .5 keeps
                         * its probability, every other class becomes 0 */
                        float *pred = (float *)malloc(sizeof(float) * l->classes);
                        int found = 0;
                        for (j = 0; j < l->classes; ++j) {
                            int parent = l->tree_parent[j];
                            pred[j] = p[box_index + 5 + j];
                            if (parent >= 0) pred[j] *= pred[parent];
                        }
                        for (j = l->classes - 1; j >= 0; --j) {
                            if (!found && pred[j] > .5) found = 1;
                            else pred[j] = 0;
                            cur[index].prob[j] = (scale > thresh) ? pred[j] : 0;
                        }
                        free(pred);
                    } else
                    for (j = 0; j < l->classes && j < classes; ++j) {
                        float prob = scale * p[box_index + 5 + j];
                        cur[index].prob[j] = (prob > thresh) ? prob : 0;
                    }
                }
            }
            o_correct(cur, tot, w, h, netw, neth, relative, letter);
            cur += tot;
        }
    }

    total = nboxes;
    if (nms > 0) {                                          /* do_nms_sort */
        k = total - 1;
        for (i = 0; i <= k; ++i) {
            if (dets[i].objectness == 0) {
                odet swap = dets[i];
                dets[i] = dets[k];
                dets[k] = swap;
                --k;
                --i;
            }
        }
        total = k + 1;
        for (k = 0; k < classes; ++k) {
            for (i = 0; i < total; ++i) dets[i].sort_class = k;
            qsort(dets, total, sizeof(odet), o_cmp);
            for (i = 0; i < total; ++i) {
                obox a;
                if (dets[i].prob[k] == 0) continue;
                a = dets[i].bbox;
                for (j = i + 1; j < total; ++j)
                    if (o_iou(a, dets[j].bbox) > nms) dets[j].prob[k] = 0;
            }
        }
    }

    for (i = 0; i < nboxes && i < max_rows && rows; ++i) {
        float *o = rows + (size_t)i * (6 + classes);
        o[0] = dets[i].bbox.x; o[1] = dets[i].bbox.y; o[2] = dets[i].bbox.w; o[3] = dets[i].bbox.h;
        o[4] = dets[i].objectness; o[5] = (float)dets[i].sort_class;
        for (j = 0; j < classes; ++j) o[6 + j] = dets[i].prob[j];
    }
    free(probs);
    free(dets);
    return nboxes;
}
