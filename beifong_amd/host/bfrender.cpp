// bfrender — command line front end with the reference CLI's flags
// (src/mitsuba/mitsuba.cpp:173-183): -m variant, -D name=value, -o output,
// -r (call receive() instead of render()), -v verbose.  Writes the raw
// film / ADC storage as OpenEXR (like `mitsuba scene.xml -o out.exr`, mitsuba.cpp:283-290) or, for `-o x.npy`, as a
// little-endian float32 .npy.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>

#include "render.h"

using namespace bfh;

static void save_npy(const std::string &path, const float *data, unsigned rows, unsigned cols, unsigned ch) {
    std::ofstream f(path, std::ios::binary);
    std::string hdr = "{'descr': '<f4', 'fortran_order': False, 'shape': (" + std::to_string(rows) + ", " +
                      std::to_string(cols) + ", " + std::to_string(ch) + "), }";
    size_t total = 10 + hdr.size() + 1;
    hdr.append((64 - total % 64) % 64, ' ');
    hdr.push_back('\n');
    unsigned short hl = (unsigned short) hdr.size();
    f.write("\x93NUMPY\x01\x00", 8);
    f.write((const char *) &hl, 2);
    f.write(hdr.data(), (std::streamsize) hdr.size());
    f.write((const char *) data, (std::streamsize) ((size_t) rows * cols * ch * 4));
}

int main(int argc, char **argv) {
    // options of src/mitsuba/mitsuba.cpp:171-183 (short and long forms); --gpus is ours
    std::string variant_name = "scalar_rgb", output;
    std::vector<std::string> scene_files;
    bool do_receive = false;
    int n_gpus = 1;
    size_t endpoint_i = 0;                      // -s: index into scene->sensors() (with -r: scene->receivers())
    xml::ParameterList params;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto is = [&](const char *s, const char *l) { return a == s || a == l; };
        if (is("-m", "--mode") && i + 1 < argc) variant_name = argv[++i];
        else if (is("-o", "--output") && i + 1 < argc) output = argv[++i];
        else if (is("-r", "--receiver")) do_receive = true;
        else if (is("-s", "--sensor") && i + 1 < argc) endpoint_i = (size_t) std::max(0, atoi(argv[++i]));
        else if (is("-v", "--verbose")) set_log_level(Debug);
        else if (is("-u", "--update")) {}                                    // scene version upgrade: the loader takes any version
        else if (is("-t", "--threads") && i + 1 < argc) ++i;                 // thread count: the work runs on the GPU
        else if (a == "-a" && i + 1 < argc) ++i;                             // resource search paths: files resolve against the scene's directory
        else if (a == "--gpus" && i + 1 < argc) n_gpus = atoi(argv[++i]);   // sample shards over N GPUs + RCCL all-reduce (ours)
        else if (a.rfind("-D", 0) == 0 || a == "--define") {
            std::string kv = (a != "--define" && a.size() > 2) ? a.substr(2) : (i + 1 < argc ? argv[++i] : "");
            size_t k = kv.find('=');
            if (k == std::string::npos) {
                fprintf(stderr, "-D expects name=value\n");
                return 2;
            }
            params.emplace_back(kv.substr(0, k), kv.substr(k + 1));
        } else if (is("-h", "--help")) {
            printf("usage: bfrender [-m variant] [-D name=value]... [-s index] [-r] [--gpus N] [-o out.exr|out.npy] [-v] scene.xml...\n");
            return 0;
        } else {
            scene_files.push_back(a);
        }
    }
    if (scene_files.empty()) {
        fprintf(stderr, "bfrender: no scene file given (try --help)\n");
        return 2;
    }
    for (const std::string &scene_file : scene_files) {
    try {
        set_variant(variant_name);
        set_gpu_count(n_gpus);
        ref<Object> obj = xml::load_file(scene_file, params);
        auto *scene = dynamic_cast<Scene *>(obj.get());
        if (!scene) Throw("top-level object is not a scene");
        Integrator *in = scene->integrator();
        const float *data;
        unsigned rows, cols, ch;
        const bf_stats *st;
        const std::vector<std::string> *names;
        if (do_receive) {
            if (scene->receivers().empty()) Throw("-r given but the scene has no receiver");
            if (endpoint_i >= scene->receivers().size()) Throw("Specified sensor index is out of bounds!");
            Receiver *r = scene->receivers()[endpoint_i].get();
            in->receive(scene, r);
            data = r->adc()->bitmap().data();
            rows = r->adc()->window_f_bins();
            cols = r->adc()->window_t_bins();
            ch = (unsigned) r->adc()->channels().size();
            names = &r->adc()->channels();
        } else {
            if (scene->sensors().empty()) Throw("the scene has no sensor");
            if (endpoint_i >= scene->sensors().size()) Throw("Specified sensor index is out of bounds!");
            Sensor *s = scene->sensors()[endpoint_i].get();
            in->render(scene, s);
            data = s->film()->bitmap().data();
            rows = s->film()->height();
            cols = s->film()->width();
            ch = (unsigned) s->film()->channels().size();
            names = &s->film()->channels();
        }
        st = &in->last_stats().stats;
        printf("rendered %llu paths, %llu rays in %.3f ms (kernels %.3f ms) -> [%u, %u, %u]\n",
               (unsigned long long) st->n_paths, (unsigned long long) (st->n_rays_closest + st->n_rays_shadow),
               in->last_stats().wall_ms, st->kernel_ms, rows, cols, ch);
        std::string out = output;
        if (out.empty()) {
            size_t k = scene_file.find_last_of('.');
            out = (k == std::string::npos ? scene_file : scene_file.substr(0, k)) + ".exr";
        }
        if (out.size() > 4 && out.compare(out.size() - 4, 4, ".npy") == 0)
            save_npy(out, data, rows, cols, ch);
        else
            write_exr(out, cols, rows, *names, data);
        printf("wrote %s\n", out.c_str());
    } catch (const std::exception &e) {
        fprintf(stderr, "bfrender: %s\n", e.what());
        return 1;
    }
    }
    return 0;
}
