// uvo_config.cpp -- YAML / parameter-tree loading for the uvo_libraries surface (see uvo_config.h).
#include "uvo_libraries_hip/uvo_config.h"

#include <cmath>
#include <cstdlib>
#include <fstream>
#include <sstream>

using namespace uvocv;

Mat R_left, t_left, R_right, t_right;
double fx, fy, ccx, ccy, k1, k2, p1, p2;
double fx_left, fy_left, ccx_left, ccy_left, fx_right, fy_right, ccx_right, ccy_right;
double k1_left, k2_left, p1_left, p2_left, k1_right, k2_right, p1_right, p2_right;
int    NODE_FREQ = 20;
int    FPS = 100;  bool SHOW_MATCHES = false;

namespace uvo_hip {

namespace {

std::string trim(const std::string& s)
{
    size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
    return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}
// a '#' starts a comment unless it sits inside quotes
std::string strip_comment(const std::string& line)
{
    char q = 0;
    for (size_t i = 0; i < line.size(); i++) {
        const char c = line[i];
        if (q) { if (c == q) q = 0; }
        else if (c == '\'' || c == '"') q = c;
        else if (c == '#' && (i == 0 || line[i - 1] == ' ' || line[i - 1] == '\t')) return line.substr(0, i);
    }
    return line;
}
bool parse_number(const std::string& t, ParamTree::Value& v)
{
    if (t.empty()) return false;
    char* end = nullptr;
    const long long i = strtoll(t.c_str(), &end, 10);
    if (*end == 0) { v.kind = ParamTree::Value::kInt; v.i = i; v.d = (double)i; return true; }
    const double d = strtod(t.c_str(), &end);
    if (*end == 0) { v.kind = ParamTree::Value::kDouble; v.d = d; return true; }
    return false;
}
ParamTree::Value parse_scalar(const std::string& raw)
{
    ParamTree::Value v;
    const std::string t = trim(raw);
    if (t.size() >= 2 && (t.front() == '\'' || t.front() == '"') && t.back() == t.front()) { v.kind = ParamTree::Value::kString; v.s = t.substr(1, t.size() - 2); return v; }
    if (t == "true" || t == "True" || t == "TRUE") { v.kind = ParamTree::Value::kBool; v.b = true; return v; }
    if (t == "false" || t == "False" || t == "FALSE") { v.kind = ParamTree::Value::kBool; v.b = false; return v; }
    if (t.size() >= 2 && t.front() == '[' && t.back() == ']') {
        v.kind = ParamTree::Value::kList;
        std::stringstream ss(t.substr(1, t.size() - 2)); std::string item;
        while (std::getline(ss, item, ',')) { ParamTree::Value e; if (parse_number(trim(item), e)) v.list.push_back(e.d); }
        return v;
    }
    if (parse_number(t, v)) return v;
    v.kind = ParamTree::Value::kString; v.s = t;
    return v;
}

}  // namespace

void ParamTree::load_yaml_text(const std::string& text, const std::string& ns)
{
    std::vector<std::pair<int, std::string>> stack;          // (indent, key) of the enclosing mappings
    std::stringstream ss(text); std::string line;
    while (std::getline(ss, line)) {
        line = strip_comment(line);
        if (trim(line).empty()) continue;
        const int indent = (int)line.find_first_not_of(" \t");
        const std::string body = trim(line);
        const size_t colon = body.find(':');
        if (colon == std::string::npos) continue;            // not a mapping entry: nothing the node reads
        const std::string key = trim(body.substr(0, colon)), rest = trim(body.substr(colon + 1));
        while (!stack.empty() && stack.back().first >= indent) stack.pop_back();
        if (rest.empty()) { stack.push_back({indent, key}); continue; }
        std::string full = ns;
        for (const auto& e : stack) full += "/" + e.second;
        full += "/" + key;
        values_[full] = parse_scalar(rest);
    }
}

void ParamTree::load_yaml_file(const std::string& path, const std::string& ns)
{
    std::ifstream f(path);
    if (!f) throw Error(UVO_INVALID_ARG, "cannot read parameter file " + path);
    std::stringstream ss; ss << f.rdbuf();
    load_yaml_text(ss.str(), ns);
}

bool ParamTree::getParam(const std::string& key, int& out) const
{
    auto it = values_.find(key);
    if (it == values_.end()) return false;
    const Value& v = it->second;
    if (v.kind == Value::kInt) { out = (int)v.i; return true; }
    if (v.kind == Value::kDouble) {                          // ros::param::getImpl(int): fmod(d, 1.0) < 0.5 ? floor : ceil
        double d = v.d;
        d = std::fmod(d, 1.0) < 0.5 ? std::floor(d) : std::ceil(d);
        out = (int)d; return true;
    }
    return false;
}
bool ParamTree::getParam(const std::string& key, double& out) const
{
    auto it = values_.find(key);
    if (it == values_.end()) return false;
    if (it->second.kind == Value::kInt || it->second.kind == Value::kDouble) { out = it->second.d; return true; }
    return false;
}
bool ParamTree::getParam(const std::string& key, bool& out) const
{
    auto it = values_.find(key);
    if (it == values_.end() || it->second.kind != Value::kBool) return false;
    out = it->second.b; return true;
}
bool ParamTree::getParam(const std::string& key, std::string& out) const
{
    auto it = values_.find(key);
    if (it == values_.end() || it->second.kind != Value::kString) return false;
    out = it->second.s; return true;
}
bool ParamTree::getParam(const std::string& key, std::vector<double>& out) const
{
    auto it = values_.find(key);
    if (it == values_.end() || it->second.kind != Value::kList) return false;
    out = it->second.list; return true;
}

}  // namespace uvo_hip

using uvo_hip::ParamTree;

void get_VO_parameters(const ParamTree& node_obj)
{
    node_obj.getParam("/node_freq", NODE_FREQ);
    node_obj.getParam("/preprocessing/desired_width", DESIRED_WIDTH);
    node_obj.getParam("/preprocessing/clahe", CLAHE_CORRECTION);
    node_obj.getParam("/preprocessing/clip_limit", CLIP_LIMIT);
    node_obj.getParam("/vo_params/distance", DISTANCE);
    node_obj.getParam("/vo_params/feature_detector", FEATURE_DETECTOR);
    node_obj.getParam("/vo_params/lowe_ratio_test", LOWE_RATIO_THRESHOLD);
    node_obj.getParam("/vo_params/essential_outlier_method", ESSENTIAL_OUTLIER_METHOD);
    node_obj.getParam("/vo_params/essential_max_iters", ESSENTIAL_MAX_ITERS);
    node_obj.getParam("/vo_params/essential_confidence", ESSENTIAL_CONFIDENCE);
    node_obj.getParam("/vo_params/essential_threshold", ESSENTIAL_THRESHOLD);
    node_obj.getParam("/vo_params/homography_outlier_method", HOMOGRAPHY_OUTLIER_METHOD);
    node_obj.getParam("/vo_params/homography_max_iters", HOMOGRAPHY_MAX_ITERS);
    node_obj.getParam("/vo_params/homography_confidence", HOMOGRAPHY_CONFIDENCE);
    node_obj.getParam("/vo_params/homography_threshold", HOMOGRAPHY_THRESHOLD);
    node_obj.getParam("/vo_params/homography_distance", HOMOGRAPHY_DISTANCE);
    node_obj.getParam("/vo_params/valid_point_fraction", VPF_THRESHOLD);
    node_obj.getParam("/vo_params/reprojection_threshold", REPROJECTION_TOLERANCE);
    node_obj.getParam("/vo_params/min_num_features", MIN_NUM_FEATURES);
    node_obj.getParam("/vo_params/min_num_3Dpoints", MIN_NUM_3DPOINTS);
    node_obj.getParam("/vo_params/min_num_inliers", MIN_NUM_INLIERS);
    node_obj.getParam("/vo_params/iterations_count", ITERATIONS_COUNT);
    node_obj.getParam("/vo_params/reprojection_error", REPROJECTION_ERROR_THRESHOLD);
    node_obj.getParam("/vo_params/confidence", CONFIDENCE);
    node_obj.getParam("/vo_params/use_extrinsic_guess", USE_EXTRINSIC_GUESS);
    node_obj.getParam("/vo_params/pnp_method_flag", PNP_METHOD_FLAG);
    node_obj.getParam("/visualization/fps", FPS);
    node_obj.getParam("/visualization/show_match", SHOW_MATCHES);
    node_obj.getParam("/surf_params/min_hessian", SURF_MIN_HESSIAN);
    node_obj.getParam("/surf_params/n_octaves", SURF_OCTAVES_NUMBER);
    node_obj.getParam("/surf_params/n_octave_layers", SURF_OCTAVES_LAYERS);
    node_obj.getParam("/surf_params/extended", SURF_EXTENDED);
    node_obj.getParam("/surf_params/upright", SURF_UPRIGHT);
}

void get_mono_camera_parameters(const ParamTree& node_obj, std::string CAMERA_NAME)
{
    node_obj.getParam("/" + CAMERA_NAME + "/camera_intrinsic/fx", fx);
    node_obj.getParam("/" + CAMERA_NAME + "/camera_intrinsic/fy", fy);
    node_obj.getParam("/" + CAMERA_NAME + "/camera_intrinsic/ccx", ccx);
    node_obj.getParam("/" + CAMERA_NAME + "/camera_intrinsic/ccy", ccy);
    node_obj.getParam("/" + CAMERA_NAME + "/distortion_coefficient/radial/k1", k1);
    node_obj.getParam("/" + CAMERA_NAME + "/distortion_coefficient/radial/k2", k2);
    node_obj.getParam("/" + CAMERA_NAME + "/distortion_coefficient/tangential/p1", p1);
    node_obj.getParam("/" + CAMERA_NAME + "/distortion_coefficient/tangential/p2", p2);
}

static void read_matrix(const ParamTree& node_obj, const std::string& base, Mat& out)
{
    std::vector<double> data; int rows = 0, cols = 0;
    if (!node_obj.getParam(base + "/data", data) || !node_obj.getParam(base + "/rows", rows) || !node_obj.getParam(base + "/cols", cols)) return;
    if (rows <= 0 || cols <= 0 || (size_t)rows * cols != data.size()) return;
    Mat m(rows, cols, CV_64FC1);
    for (int i = 0; i < rows * cols; i++) m.at<double>(i / cols, i % cols) = data[i];
    out = m;
}

void get_stereo_camera_parameters(const ParamTree& node_obj, std::string CAMERA_NAME)
{
    const std::string c = "/" + CAMERA_NAME;
    node_obj.getParam(c + "/camera_intrinsic_left/fx", fx_left);   node_obj.getParam(c + "/camera_intrinsic_left/fy", fy_left);
    node_obj.getParam(c + "/camera_intrinsic_left/ccx", ccx_left); node_obj.getParam(c + "/camera_intrinsic_left/ccy", ccy_left);
    node_obj.getParam(c + "/camera_intrinsic_right/fx", fx_right);   node_obj.getParam(c + "/camera_intrinsic_right/fy", fy_right);
    node_obj.getParam(c + "/camera_intrinsic_right/ccx", ccx_right); node_obj.getParam(c + "/camera_intrinsic_right/ccy", ccy_right);
    node_obj.getParam(c + "/distortion_coefficient_left/radial/k1", k1_left);     node_obj.getParam(c + "/distortion_coefficient_left/radial/k2", k2_left);
    node_obj.getParam(c + "/distortion_coefficient_left/tangential/p1", p1_left); node_obj.getParam(c + "/distortion_coefficient_left/tangential/p2", p2_left);
    node_obj.getParam(c + "/distortion_coefficient_right/radial/k1", k1_right);     node_obj.getParam(c + "/distortion_coefficient_right/radial/k2", k2_right);
    node_obj.getParam(c + "/distortion_coefficient_right/tangential/p1", p1_right); node_obj.getParam(c + "/distortion_coefficient_right/tangential/p2", p2_right);
    read_matrix(node_obj, c + "/left_camera_rotation_matrix", R_left);
    read_matrix(node_obj, c + "/left_camera_translation_vector", t_left);
    read_matrix(node_obj, c + "/right_camera_rotation_matrix", R_right);
    read_matrix(node_obj, c + "/right_camera_translation_vector", t_right);
}
