// host_detect.cpp -- detection decode + per-class sort-NMS on the host, from
// the YOLO / REGION head outputs that yl_network_predict pulled back.
//
// Behavioural mirror of (types and evaluation order kept so results are
// identical to the reference on the same head tensors):
//   get_network_boxes / make_network_boxes / num_detections   src/additionally.c:4403 / 4238 / 4222
//   yolo_num_detections / get_yolo_detections / get_yolo_box   src/additionally.c:4207 / 4328 / 4317
//   correct_yolo_boxes                                         src/additionally.c:4281
//   custom_get_region_detections                               src/additionally.c:4363
//   get_region_boxes_cpu / get_region_box_cpu                  src/yolov2_forward_network.c:664 / 653
//   do_nms_sort / nms_comparator / box_iou                     src/box.c:296 / 280 / 94
// Extension: `image` selects the batch item (the reference is hard-wired to item 0).
#include "yl_internal.h"

#include <cmath>
#include <cstdlib>
#include <cstring>

namespace yl {

namespace {

struct Box { float x, y, w, h; };

// same size/field order as the reference's `detection` (src/box.h:9-17) so qsort
// sees identical element geometry
struct Det {
    Box bbox;
    int classes;
    float *prob;
    float *mask;
    float objectness;
    int sort_class;
};

inline int entry_index(const Layer &l, int batch, int location, int entry) {
    const int n = location / (l.w * l.h);
    const int loc = location % (l.w * l.h);
    return batch * l.outputs + n * l.w * l.h * (4 + l.classes + 1) + entry * l.w * l.h + loc;
}

float overlap(float x1, float w1, float x2, float w2) {
    float l1 = x1 - w1 / 2;
    float l2 = x2 - w2 / 2;
    float left = l1 > l2 ? l1 : l2;
    float r1 = x1 + w1 / 2;
    float r2 = x2 + w2 / 2;
    float right = r1 < r2 ? r1 : r2;
    return right - left;
}

float box_intersection(Box a, Box b) {
    float w = overlap(a.x, a.w, b.x, b.w);
    float h = overlap(a.y, a.h, b.y, b.h);
    if (w < 0 || h < 0) return 0;
    return w * h;
}

float box_union(Box a, Box b) {
    float i = box_intersection(a, b);
    return a.w * a.h + b.w * b.h - i;
}

float box_iou(Box a, Box b) { return box_intersection(a, b) / box_union(a, b); }

int nms_comparator(const void *pa, const void *pb) {
    const Det &a = *(const Det *)pa;
    const Det &b = *(const Det *)pb;
    float diff = 0;
    if (b.sort_class >= 0) diff = a.prob[b.sort_class] - b.prob[b.sort_class];
    else diff = a.objectness - b.objectness;
    if (diff < 0) return 1;
    else if (diff > 0) return -1;
    return 0;
}

void correct_boxes(Det *dets, int n, int w, int h, int netw, int neth, int relative, int letter) {
    int new_w = 0, new_h = 0;
    if (letter) {
        if (((float)netw / w) < ((float)neth / h)) { new_w = netw; new_h = (h * netw) / w; }
        else { new_h = neth; new_w = (w * neth) / h; }
    } else { new_w = netw; new_h = neth; }
    for (int i = 0; i < n; ++i) {
        Box b = dets[i].bbox;
        b.x = (b.x - (netw - new_w) / 2. / netw) / ((float)new_w / netw);
        b.y = (b.y - (neth - new_h) / 2. / neth) / ((float)new_h / neth);
        b.w *= (float)netw / new_w;
        b.h *= (float)neth / new_h;
        if (!relative) { b.x *= w; b.w *= w; b.y *= h; b.h *= h; }
        dets[i].bbox = b;
    }
}

}  // namespace

int get_boxes_host(Network &net, int image, int w, int h, float thresh, int relative,
                   int letter, float nms, float *rows, int max_rows, int *classes_out) {
    if (image < 0 || image >= net.batch) { set_error("image index out of range"); return YL_ERR_ARG; }
    const Layer &last = net.layers.back();
    const int classes = last.classes;
    if (classes_out) *classes_out = classes;

    // num_detections
    int nboxes = 0;
    for (const Layer &l : net.layers) {
        if (l.type == YL_YOLO) {
            if (!l.host_output) { set_error("head outputs not on host: call yl_network_predict/pull_heads first"); return YL_ERR_STATE; }
            for (int i = 0; i < l.w * l.h; ++i)
                for (int n = 0; n < l.n; ++n)
                    if (l.host_output[entry_index(l, image, n * l.w * l.h + i, 4)] > thresh) ++nboxes;
        } else if (l.type == YL_REGION) {
            if (!l.host_output) { set_error("head outputs not on host: call yl_network_predict/pull_heads first"); return YL_ERR_STATE; }
            nboxes += l.w * l.h * l.n;
        }
    }
    std::vector<Det> dets(nboxes);
    std::vector<float> probs((size_t)nboxes * (classes > 0 ? classes : 1), 0.f);
    for (int i = 0; i < nboxes; ++i) {
        memset(&dets[i], 0, sizeof(Det));
        dets[i].prob = probs.data() + (size_t)i * classes;
    }

    Det *cur = dets.data();
    for (const Layer &l : net.layers) {
        if (l.type == YL_YOLO) {
            const float *predictions = l.host_output;
            int count = 0;
            const int lwh = l.w * l.h;
            for (int i = 0; i < lwh; ++i) {
                const int row = i / l.w, col = i % l.w;
                for (int n = 0; n < l.n; ++n) {
                    const int obj_index = entry_index(l, image, n * lwh + i, 4);
                    const float objectness = predictions[obj_index];
                    if (objectness > thresh) {
                        const int box_index = entry_index(l, image, n * lwh + i, 0);
                        const int an = l.mask[n];
                        Box b;
                        b.x = (col + predictions[box_index + 0 * lwh]) / l.w;
                        b.y = (row + predictions[box_index + 1 * lwh]) / l.h;
                        b.w = exp((double)predictions[box_index + 2 * lwh]) * l.anchors[2 * an] / net.w;
                        b.h = exp((double)predictions[box_index + 3 * lwh]) * l.anchors[2 * an + 1] / net.h;
                        cur[count].bbox = b;
                        cur[count].objectness = objectness;
                        cur[count].classes = l.classes;
                        for (int j = 0; j < l.classes && j < classes; ++j) {
                            const int class_index = entry_index(l, image, n * lwh + i, 4 + 1 + j);
                            const float prob = objectness * predictions[class_index];
                            cur[count].prob[j] = (prob > thresh) ? prob : 0;
                        }
                        ++count;
                    }
                }
            }
            correct_boxes(cur, count, w, h, net.w, net.h, relative, letter);
            cur += count;
        } else if (l.type == YL_REGION) {
            const float *predictions = l.host_output + (size_t)image * l.outputs;
            const int total = l.w * l.h * l.n;
            for (int i = 0; i < l.w * l.h; ++i) {
                const int row = i / l.w, col = i % l.w;
                for (int n = 0; n < l.n; ++n) {
                    const int index = i * l.n + n;
                    const int p_index = index * (l.classes + 5) + 4;
                    const float scale = predictions[p_index];
                    const int box_index = index * (l.classes + 5);
                    Box b;
                    // logistic_activate(x) = 1./(1. + exp(-x)) evaluated in double, returned as float
                    const float lx = (float)(1. / (1. + exp((double)(-predictions[box_index + 0]))));
                    const float ly = (float)(1. / (1. + exp((double)(-predictions[box_index + 1]))));
                    b.x = (col + lx) / l.w;
                    b.y = (row + ly) / l.h;
                    b.w = expf(predictions[box_index + 2]) * l.anchors[2 * n] / l.w;
                    b.h = expf(predictions[box_index + 3]) * l.anchors[2 * n + 1] / l.h;
                    // get_region_boxes_cpu is called with w = h = 1: `boxes[index].x *= w`
                    b.x *= 1; b.y *= 1; b.w *= 1; b.h *= 1;
                    const int class_index = index * (l.classes + 5) + 5;
                    cur[index].classes = l.classes;
                    cur[index].bbox = b;
                    cur[index].objectness = 1;
                    for (int j = 0; j < l.classes && j < classes; ++j) {
                        const float prob = scale * predictions[class_index + j];
                        cur[index].prob[j] = (prob > thresh) ? prob : 0;
                    }
                }
            }
            correct_boxes(cur, total, w, h, net.w, net.h, relative, letter);
            cur += total;
        }
    }

    int total = nboxes;
    if (nms > 0) {
        int k = total - 1;
        for (int i = 0; i <= k; ++i) {
            if (dets[i].objectness == 0) {
                Det swap = dets[i];
                dets[i] = dets[k];
                dets[k] = swap;
                --k;
                --i;
            }
        }
        total = k + 1;
        for (k = 0; k < classes; ++k) {
            for (int i = 0; i < total; ++i) dets[i].sort_class = k;
            qsort(dets.data(), total, sizeof(Det), nms_comparator);
            for (int i = 0; i < total; ++i) {
                if (dets[i].prob[k] == 0) continue;
                const Box a = dets[i].bbox;
                for (int j = i + 1; j < total; ++j) {
                    if (box_iou(a, dets[j].bbox) > nms) dets[j].prob[k] = 0;
                }
            }
        }
    }

    const int stride = 6 + classes;
    const int nrows = nboxes < max_rows ? nboxes : max_rows;
    for (int i = 0; i < nrows && rows; ++i) {
        float *o = rows + (size_t)i * stride;
        o[0] = dets[i].bbox.x; o[1] = dets[i].bbox.y; o[2] = dets[i].bbox.w; o[3] = dets[i].bbox.h;
        o[4] = dets[i].objectness; o[5] = (float)dets[i].sort_class;
        for (int j = 0; j < classes; ++j) o[6 + j] = dets[i].prob[j];
    }
    return nboxes;
}

}  // namespace yl
