// fuzz_param_tree.cpp -- robustness harness of uvo_hip::ParamTree and the three loaders (tests/test_node.py builds it with
// -fsanitize=address,undefined and feeds it mutated copies of the parameter files): every case is parsed and read through the
// loaders; malformed text may be refused (uvo_hip::Error) or yield missing keys, it may not touch memory it does not own.
//   usage: fuzz_param_tree <cases.txt>      cases separated by a line "===CASE==="
#include <cstdio>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>
#include "uvo_libraries_hip/uvo_config.h"

int main(int argc, char** argv)
{
    if (argc != 2) { fprintf(stderr, "usage: %s cases.txt\n", argv[0]); return 2; }
    std::ifstream f(argv[1], std::ios::binary);
    std::stringstream ss; ss << f.rdbuf();
    const std::string all = ss.str(), sep = "===CASE===\n";
    std::vector<std::string> cases;
    for (size_t pos = 0; pos <= all.size();) {
        const size_t e = all.find(sep, pos);
        cases.push_back(all.substr(pos, e == std::string::npos ? std::string::npos : e - pos));
        if (e == std::string::npos) break;
        pos = e + sep.size();
    }
    int parsed = 0, refused = 0, loaded = 0;
    for (const std::string& text : cases) {
        uvo_hip::ParamTree t;
        try { t.load_yaml_text(text); parsed++; }
        catch (const uvo_hip::Error&) { refused++; continue; }
        int i = 0; double d = 0; bool b = false; std::string s;
        (void)t.getParam("/distance", i); (void)t.getParam("/distance", d); (void)t.getParam("/distance", b); (void)t.getParam("/distance", s);
        (void)t.getParam("/feature_detector", s); (void)t.getParam("/surf_extended", b); (void)t.getParam("/lowe_ratio_threshold", d);
        try {
            get_VO_parameters(t);
            get_mono_camera_parameters(t, "cam");
            get_stereo_camera_parameters(t, "cam");
            loaded++;
        } catch (const uvo_hip::Error&) { refused++; }
    }
    printf("FUZZ-OK cases %d parsed %d refused %d loaded %d\n", (int)cases.size(), parsed, refused, loaded);
    return 0;
}
