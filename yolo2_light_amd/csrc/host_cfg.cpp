// host_cfg.cpp -- darknet .cfg reader producing yl::Layer descriptors.
//
// Behavioural mirror (not a copy) of the reference's parser for the layer
// kinds on the hot path:
//   read_cfg / read_option            src/additionally.c:3423, 3244-3262
//   parse_net_options                 src/additionally.c:3858
//   parse_convolutional               src/additionally.c:3534  (pad -> size/2, :3539-3541)
//   parse_maxpool                     src/additionally.c:3701  (padding default size-1)
//   parse_route / parse_shortcut      src/additionally.c:3764 / 3746
//   parse_upsample / parse_reorg      src/additionally.c:3736 / 3719
//   parse_yolo / parse_region         src/additionally.c:3646 / 3573
//   make_*_layer output geometry      src/additionally.c:2299-2910
// Only geometry and hyper-parameters are produced here; buffers live on the GPU.
#include "yl_internal.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <sstream>

namespace yl {

namespace {

struct Section {
    std::string type;
    std::vector<std::pair<std::string, std::string>> opts;
    const std::string *find(const char *key) const {
        for (auto &kv : opts) if (kv.first == key) return &kv.second;
        return nullptr;
    }
    int geti(const char *key, int def) const {
        const std::string *v = find(key);
        return v ? atoi(v->c_str()) : def;
    }
    float getf(const char *key, float def) const {
        const std::string *v = find(key);
        return v ? (float)atof(v->c_str()) : def;
    }
    std::string gets(const char *key, const char *def) const {
        const std::string *v = find(key);
        return v ? *v : std::string(def ? def : "");
    }
};

// the reference strips ' ', '\t', '\n' (and we add '\r') from every line before use
std::string strip_ws(const std::string &s) {
    std::string o;
    o.reserve(s.size());
    for (char ch : s) if (ch != ' ' && ch != '\t' && ch != '\n' && ch != '\r') o.push_back(ch);
    return o;
}

bool read_sections(const char *path, std::vector<Section> &secs) {
    FILE *f = fopen(path, "r");
    if (!f) { set_error(std::string("cannot open cfg file: ") + path); return false; }
    std::string line;
    int ch;
    auto flush = [&]() -> bool {
        std::string s = strip_ws(line);
        line.clear();
        if (s.empty()) return true;
        if (s[0] == '[') { Section sec; sec.type = s; secs.push_back(sec); return true; }
        if (s[0] == '#' || s[0] == ';') return true;
        size_t eq = s.find('=');
        if (eq == std::string::npos || secs.empty()) return true;   // reference prints and skips
        secs.back().opts.emplace_back(s.substr(0, eq), s.substr(eq + 1));
        return true;
    };
    while ((ch = fgetc(f)) != EOF) {
        if (ch == '\n') flush(); else line.push_back((char)ch);
    }
    flush();
    fclose(f);
    return true;
}

std::vector<float> parse_float_list(const std::string &s) {
    std::vector<float> v;
    size_t pos = 0;
    while (pos <= s.size()) {
        size_t comma = s.find(',', pos);
        std::string tok = s.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos);
        v.push_back((float)atof(tok.c_str()));
        if (comma == std::string::npos) break;
        pos = comma + 1;
    }
    return v;
}

std::vector<int> parse_int_list(const std::string &s) {
    std::vector<int> v;
    size_t pos = 0;
    while (pos <= s.size()) {
        size_t comma = s.find(',', pos);
        std::string tok = s.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos);
        v.push_back(atoi(tok.c_str()));
        if (comma == std::string::npos) break;
        pos = comma + 1;
    }
    return v;
}

int activation_from(const std::string &s, bool &ok) {
    ok = true;
    // get_activation (src/additionally.h:107-125); an unknown name is an error here (the reference falls back to ReLU
    // with a message on stderr)
    static const struct { const char *name; int id; } table[] = {
        {"logistic", YL_LOGISTIC}, {"loggy", YL_LOGGY}, {"relu", YL_RELU}, {"elu", YL_ELU}, {"relie", YL_RELIE},
        {"plse", YL_PLSE}, {"hardtan", YL_HARDTAN}, {"lhtan", YL_LHTAN}, {"linear", YL_LINEAR}, {"ramp", YL_RAMP},
        {"leaky", YL_LEAKY}, {"tanh", YL_TANH}, {"stair", YL_STAIR},
    };
    for (const auto &e : table)
        if (s == e.name) return e.id;
    ok = false;
    return YL_LINEAR;
}

}  // namespace

// read_tree (src/additionally.c:1895-1945): one "name parent" line per class; a new softmax group starts wherever
// the parent changes.  Every line counts as a node (sscanf on a blank line leaves parent = -1), like the reference.
static int read_tree_file(const std::string &path, std::vector<int> &parent, std::vector<int> &group_size)
{
    FILE *fp = fopen(path.c_str(), "r");
    if (!fp) return YL_ERR_IO;
    parent.clear(); group_size.clear();
    int last_parent = -1, gsize = 0;
    char *line = nullptr;
    size_t cap = 0;
    ssize_t len;
    while ((len = getline(&line, &cap, fp)) >= 0) {
        char id[256];
        int par = -1;
        sscanf(line, "%255s %d", id, &par);
        parent.push_back(par);
        if (par != last_parent) {
            group_size.push_back(gsize);
            gsize = 0;
            last_parent = par;
        }
        ++gsize;
    }
    free(line);
    fclose(fp);
    group_size.push_back(gsize);
    return YL_OK;
}

int parse_cfg_file(const char *path, int batch, int quantized, Network &net) {
    std::vector<Section> secs;
    if (!read_sections(path, secs)) return YL_ERR_IO;
    if (secs.empty()) { set_error("Config file has no sections"); return YL_ERR_CFG; }
    const Section &ns = secs[0];
    if (ns.type != "[net]" && ns.type != "[network]") {
        set_error("First section must be [net] or [network]");
        return YL_ERR_CFG;
    }
    int cfg_batch = ns.geti("batch", 1);
    int subdivs = ns.geti("subdivisions", 1);
    int time_steps = ns.geti("time_steps", 1);
    cfg_batch /= (subdivs > 0 ? subdivs : 1);
    cfg_batch *= time_steps;
    net.batch = batch > 0 ? batch : cfg_batch;
    net.h = ns.geti("height", 0);
    net.w = ns.geti("width", 0);
    net.c = ns.geti("channels", 0);
    net.quantized = quantized;
    if (!(net.h && net.w && net.c)) { set_error("No input parameters supplied"); return YL_ERR_CFG; }
    if (const std::string *ic = ns.find("input_calibration")) net.input_calibration = parse_float_list(*ic);

    int ph = net.h, pw = net.w, pc = net.c, pinputs = net.h * net.w * net.c;
    // the parser's running `params.quantized` (src/additionally.c:3996-4004): switched off for good by
    // the first convolution that has a [yolo] section two sections further on
    int params_quantized = quantized;
    net.layers.clear();
    for (size_t si = 1; si < secs.size(); ++si) {
        const Section &s = secs[si];
        const int idx = (int)si - 1;
        Layer l;
        l.batch = net.batch;
        char where[64];
        snprintf(where, sizeof(where), "layer %d %s: ", idx, s.type.c_str());

        if (s.type == "[convolutional]" || s.type == "[conv]") {
            l.type = YL_CONVOLUTIONAL;
            l.n = s.geti("filters", 1);
            l.size = s.geti("size", 1);
            l.stride = s.geti("stride", 1);
            if (l.size <= 0 || l.stride <= 0 || l.n <= 0) { set_error(std::string(where) + "filters, size and stride must be positive"); return YL_ERR_CFG; }
            int pad = s.geti("pad", 0);
            int padding = s.geti("padding", 0);
            if (pad) padding = l.size / 2;
            l.pad = padding;
            bool ok;
            l.activation = activation_from(s.gets("activation", "logistic"), ok);
            if (!ok) { set_error(std::string(where) + "unsupported activation " + s.gets("activation", "")); return YL_ERR_UNSUPPORTED; }
            l.h = ph; l.w = pw; l.c = pc;
            if (!(l.h && l.w && l.c)) { set_error(std::string(where) + "Layer before convolutional layer must output image."); return YL_ERR_CFG; }
            l.batch_normalize = s.geti("batch_normalize", 0);
            l.xnor = s.geti("xnor", 0);
            // l.quantized of the reference's parser = the layer set its GPU path quantises
            // (parse_network_cfg src/additionally.c:3996-4004, parse_convolutional :3557-3559)
            if (si + 2 < secs.size() && secs[si + 2].type == "[yolo]") params_quantized = 0;
            l.gpu_quantized = params_quantized;
            if (idx == 0 || l.activation == YL_LINEAR || (idx > 1 && l.stride > 1) || l.size == 1) l.gpu_quantized = 0;
            if (s.geti("binary", 0)) { set_error(std::string(where) + "binary=1 is not on the hot path"); return YL_ERR_UNSUPPORTED; }
            l.out_h = (l.h + 2 * l.pad - l.size) / l.stride + 1;
            l.out_w = (l.w + 2 * l.pad - l.size) / l.stride + 1;
            l.out_c = l.n;
            l.outputs = l.out_h * l.out_w * l.out_c;
            l.inputs = l.w * l.h * l.c;
            const size_t nw = (size_t)l.n * l.c * l.size * l.size;
            l.weights.assign(nw, 0.f);
            l.biases.assign(l.n, 0.f);
            if (l.batch_normalize) {
                l.scales.assign(l.n, 1.f);
                l.rolling_mean.assign(l.n, 0.f);
                l.rolling_variance.assign(l.n, 0.f);
            }
        } else if (s.type == "[maxpool]" || s.type == "[max]") {
            l.type = YL_MAXPOOL;
            l.stride = s.geti("stride", 1);
            l.size = s.geti("size", l.stride);
            if (l.size <= 0 || l.stride <= 0) { set_error(std::string(where) + "size and stride must be positive"); return YL_ERR_CFG; }
            l.pad = s.geti("padding", l.size - 1);
            l.h = ph; l.w = pw; l.c = pc;
            if (!(l.h && l.w && l.c)) { set_error(std::string(where) + "Layer before maxpool layer must output image."); return YL_ERR_CFG; }
            l.out_w = (l.w + l.pad - l.size) / l.stride + 1;
            l.out_h = (l.h + l.pad - l.size) / l.stride + 1;
            l.out_c = l.c;
            l.outputs = l.out_h * l.out_w * l.out_c;
            l.inputs = l.h * l.w * l.c;
        } else if (s.type == "[route]") {
            l.type = YL_ROUTE;
            const std::string *ls = s.find("layers");
            if (!ls) { set_error(std::string(where) + "Route Layer must specify input layers"); return YL_ERR_CFG; }
            std::vector<int> ids = parse_int_list(*ls);
            l.n = (int)ids.size();
            int outputs = 0;
            for (int &id : ids) {
                if (id < 0) id = idx + id;
                if (id < 0 || id >= idx) { set_error(std::string(where) + "route index out of range"); return YL_ERR_CFG; }
                l.input_layers.push_back(id);
                l.input_sizes.push_back(net.layers[id].outputs);
                outputs += net.layers[id].outputs;
            }
            l.outputs = outputs;
            l.inputs = outputs;
            const Layer &first = net.layers[l.input_layers[0]];
            l.out_w = first.out_w; l.out_h = first.out_h; l.out_c = first.out_c;
            for (int i = 1; i < l.n; ++i) {
                const Layer &nx = net.layers[l.input_layers[i]];
                if (nx.out_w == first.out_w && nx.out_h == first.out_h) l.out_c += nx.out_c;
                else l.out_h = l.out_w = l.out_c = 0;
            }
            l.w = l.out_w; l.h = l.out_h; l.c = l.out_c;
        } else if (s.type == "[shortcut]") {
            l.type = YL_SHORTCUT;
            const std::string *from = s.find("from");
            if (!from) { set_error(std::string(where) + "shortcut needs from="); return YL_ERR_CFG; }
            int index = atoi(from->c_str());
            if (index < 0) index = idx + index;
            if (index < 0 || index >= idx) { set_error(std::string(where) + "shortcut index out of range"); return YL_ERR_CFG; }
            const Layer &fr = net.layers[index];
            l.index = index;
            l.w = fr.out_w; l.h = fr.out_h; l.c = fr.out_c;      // dims of the added tensor
            l.out_w = pw; l.out_h = ph; l.out_c = pc;
            l.outputs = pw * ph * pc;
            l.inputs = l.outputs;
            bool ok;
            l.activation = activation_from(s.gets("activation", "linear"), ok);
            if (!ok) { set_error(std::string(where) + "unsupported activation"); return YL_ERR_UNSUPPORTED; }
        } else if (s.type == "[upsample]") {
            l.type = YL_UPSAMPLE;
            l.stride = s.geti("stride", 2);
            if (l.stride < 0) { set_error(std::string(where) + "reverse upsample is not on the hot path"); return YL_ERR_UNSUPPORTED; }
            if (l.stride == 0) { set_error(std::string(where) + "stride must be positive"); return YL_ERR_CFG; }
            l.w = pw; l.h = ph; l.c = pc;
            l.out_w = pw * l.stride; l.out_h = ph * l.stride; l.out_c = pc;
            l.outputs = l.out_w * l.out_h * l.out_c;
            l.inputs = l.w * l.h * l.c;
            l.scale = s.getf("scale", 1.f);
        } else if (s.type == "[reorg]") {
            l.type = YL_REORG;
            l.stride = s.geti("stride", 1);
            if (l.stride <= 0) { set_error(std::string(where) + "stride must be positive"); return YL_ERR_CFG; }
            if (s.geti("reverse", 0)) { set_error(std::string(where) + "reverse reorg is not on the hot path"); return YL_ERR_UNSUPPORTED; }
            l.w = pw; l.h = ph; l.c = pc;
            l.out_w = pw / l.stride; l.out_h = ph / l.stride; l.out_c = pc * l.stride * l.stride;
            l.outputs = l.out_h * l.out_w * l.out_c;
            l.inputs = l.h * l.w * l.c;
        } else if (s.type == "[yolo]") {
            l.type = YL_YOLO;
            l.classes = s.geti("classes", 20);
            l.total = s.geti("num", 1);
            l.n = l.total;
            if (const std::string *m = s.find("mask")) { l.mask = parse_int_list(*m); l.n = (int)l.mask.size(); }
            else { for (int i = 0; i < l.n; ++i) l.mask.push_back(i); }
            l.w = pw; l.h = ph; l.c = l.n * (l.classes + 4 + 1);
            l.out_w = l.w; l.out_h = l.h; l.out_c = l.c;
            l.outputs = l.h * l.w * l.n * (l.classes + 4 + 1);
            l.inputs = l.outputs;
            if (l.outputs != pinputs) {
                set_error(std::string(where) + "filters= in the [convolutional]-layer doesn't correspond to classes= or mask= in [yolo]-layer");
                return YL_ERR_CFG;
            }
            l.anchors.assign((size_t)l.total * 2, .5f);
            if (const std::string *a = s.find("anchors")) {
                std::vector<float> v = parse_float_list(*a);
                for (size_t i = 0; i < v.size() && i < (size_t)l.total * 2; ++i) l.anchors[i] = v[i];
            }
            for (int m : l.mask) if (m < 0 || m >= l.total) { set_error(std::string(where) + "mask index out of range"); return YL_ERR_CFG; }
        } else if (s.type == "[region]") {
            l.type = YL_REGION;
            l.coords = s.geti("coords", 4);
            l.classes = s.geti("classes", 20);
            l.n = s.geti("num", 1);
            l.total = l.n;
            l.softmax = s.geti("softmax", 0);
            // classfix == -1 zeroes scale < .5 in get_region_boxes_cpu (src/additionally.c:3591); no decode here honours it
            if (s.geti("classfix", 0) != 0) { set_error(std::string(where) + "classfix != 0 is not on the hot path"); return YL_ERR_UNSUPPORTED; }
            if (s.find("map") != nullptr) { set_error(std::string(where) + "map= (YOLO9000 class remapping of the evaluator) is not on the hot path"); return YL_ERR_UNSUPPORTED; }
            if (const std::string *tf = s.find("tree")) {
                const int rc = read_tree_file(*tf, l.tree_parent, l.tree_group_size);
                if (rc != YL_OK) { set_error(std::string(where) + "cannot read tree file " + *tf); return rc; }
                if ((int)l.tree_parent.size() != l.classes) { set_error(std::string(where) + "tree size != classes"); return YL_ERR_CFG; }
            }
            if (l.coords != 4) { set_error(std::string(where) + "coords != 4 unsupported"); return YL_ERR_UNSUPPORTED; }
            l.w = pw; l.h = ph; l.c = pc;
            l.outputs = l.h * l.w * l.n * (l.classes + l.coords + 1);
            l.inputs = l.outputs;
            if (l.outputs != pinputs) { set_error(std::string(where) + "region outputs != inputs"); return YL_ERR_CFG; }
            l.anchors.assign((size_t)l.n * 2, .5f);
            if (const std::string *a = s.find("anchors")) {
                std::vector<float> v = parse_float_list(*a);
                for (size_t i = 0; i < v.size() && i < (size_t)l.n * 2; ++i) l.anchors[i] = v[i];
            }
            // make_region_layer leaves out_w/out_h/out_c at 0 (src/additionally.c:2551-2570)
        } else {
            set_error(std::string(where) + "layer type is not on the hot path");
            return YL_ERR_UNSUPPORTED;
        }
        net.layers.push_back(std::move(l));
        const Layer &b = net.layers.back();
        ph = b.out_h; pw = b.out_w; pc = b.out_c; pinputs = b.outputs;
    }
    if (net.layers.empty()) { set_error("cfg has no layers"); return YL_ERR_CFG; }
    select_conv_modes(net);
    return YL_OK;
}

}  // namespace yl
