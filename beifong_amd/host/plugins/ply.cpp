// src/shapes/ply.cpp:92-470 — PLYMesh: ascii / binary_little_endian /
// binary_big_endian; vertex x y z [nx ny nz] [u v | s t | texture_u texture_v];
// face list vertex_index | vertex_indices; extra properties and elements are
// skipped; to_world applied at load time; normals recomputed when absent.
#include <cmath>
#include <cstring>
#include <fstream>
#include <sstream>

#include "../render.h"
using namespace bfh;

namespace {
struct Prop {
    std::string name;
    int type = 0, count_type = 0, item_type = 0;   // 0 = not a list
    bool list = false;
};
struct Element {
    std::string name;
    size_t count = 0;
    std::vector<Prop> props;
};
int type_id(const std::string &t) {
    static const char *names[][2] = {{"char", "int8"}, {"uchar", "uint8"}, {"short", "int16"}, {"ushort", "uint16"},
                                     {"int", "int32"}, {"uint", "uint32"}, {"float", "float32"}, {"double", "float64"}};
    for (int i = 0; i < 8; ++i)
        if (t == names[i][0] || t == names[i][1]) return i + 1;
    Throw("invalid data type \"%s\" in PLY header", t.c_str());
}
size_t type_size(int t) {
    static const size_t sz[] = {0, 1, 1, 2, 2, 4, 4, 4, 8};
    return sz[t];
}
struct Reader {
    std::istream &in;
    bool ascii, swap;
    double scalar(int t) {
        if (ascii) {
            double v;
            if (!(in >> v)) Throw("unexpected end of PLY data");
            return v;
        }
        unsigned char b[8];
        size_t n = type_size(t);
        in.read((char *) b, (std::streamsize) n);
        if ((size_t) in.gcount() != n) Throw("unexpected end of PLY data");
        if (swap)
            for (size_t i = 0; i < n / 2; ++i) std::swap(b[i], b[n - 1 - i]);
        switch (t) {
            case 1: return (double) *(int8_t *) b;
            case 2: return (double) *(uint8_t *) b;
            case 3: { int16_t v; std::memcpy(&v, b, 2); return v; }
            case 4: { uint16_t v; std::memcpy(&v, b, 2); return v; }
            case 5: { int32_t v; std::memcpy(&v, b, 4); return v; }
            case 6: { uint32_t v; std::memcpy(&v, b, 4); return v; }
            case 7: { float v; std::memcpy(&v, b, 4); return v; }
            default: { double v; std::memcpy(&v, b, 8); return v; }
        }
    }
};
}  // namespace

class PLYMesh final : public Mesh {
public:
    explicit PLYMesh(const Properties &props) : Mesh(props) {
        bool face_normals = props.bool_("face_normals", false);
        std::string path = resolve_path(props.string("filename"));
        std::ifstream in(path, std::ios::binary);
        if (!in.good()) Throw("Error while loading PLY file \"%s\": file not found!", path.c_str());
        auto fail = [&](const std::string &m) { Throw("Error while loading PLY file \"%s\": %s!", path.c_str(), m.c_str()); };
        std::string line;
        std::getline(in, line);
        if (line.substr(0, 3) != "ply") fail("invalid PLY header");
        bool ascii = false, big = false, have_format = false;
        std::vector<Element> elements;
        while (std::getline(in, line)) {
            if (!line.empty() && line.back() == '\r') line.pop_back();
            std::istringstream ss(line);
            std::string tok;
            ss >> tok;
            if (tok == "format") {
                std::string f, v;
                ss >> f >> v;
                if (f == "ascii") ascii = true;
                else if (f == "binary_little_endian") big = false;
                else if (f == "binary_big_endian") big = true;
                else fail("invalid PLY format \"" + f + "\"");
                if (v != "1.0") fail("PLY file has unknown version number \"" + v + "\"");
                have_format = true;
            } else if (tok == "comment" || tok == "obj_info") {
            } else if (tok == "element") {
                Element e;
                ss >> e.name >> e.count;
                elements.push_back(e);
            } else if (tok == "property") {
                if (elements.empty()) fail("property before element");
                Prop p;
                std::string t;
                ss >> t;
                if (t == "list") {
                    std::string ct, it;
                    ss >> ct >> it >> p.name;
                    p.list = true;
                    p.count_type = type_id(ct);
                    p.item_type = type_id(it);
                } else {
                    p.type = type_id(t);
                    ss >> p.name;
                }
                elements.back().props.push_back(p);
            } else if (tok == "end_header") {
                break;
            } else if (!tok.empty()) {
                fail("invalid token in PLY header: \"" + tok + "\"");
            }
        }
        if (!have_format) fail("PLY header lacks a format line");
        uint16_t one = 1;
        bool host_little = *(uint8_t *) &one == 1;
        Reader rd{in, ascii, !ascii && (big == host_little)};
        bool file_normals = false;
        std::vector<float> nrm;
        for (const Element &el : elements) {
            if (el.name == "vertex") {
                int ix = -1, iy = -1, iz = -1, inx = -1, iny = -1, inz = -1, iu = -1, iv = -1;
                for (size_t k = 0; k < el.props.size(); ++k) {
                    const std::string &n = el.props[k].name;
                    if (n == "x") ix = (int) k; else if (n == "y") iy = (int) k; else if (n == "z") iz = (int) k;
                    else if (n == "nx") inx = (int) k; else if (n == "ny") iny = (int) k; else if (n == "nz") inz = (int) k;
                    else if (n == "u" || n == "s" || n == "texture_u") iu = (int) k;
                    else if (n == "v" || n == "t" || n == "texture_v") iv = (int) k;
                }
                if (ix < 0 || iy < 0 || iz < 0) fail("vertex element lacks x/y/z");
                file_normals = !face_normals && inx >= 0 && iny >= 0 && inz >= 0;
                m_positions.resize(3 * el.count);
                if (file_normals) nrm.resize(3 * el.count);
                if (iu >= 0 && iv >= 0) m_texcoords.resize(2 * el.count);
                std::vector<double> vals(el.props.size());
                for (size_t i = 0; i < el.count; ++i) {
                    for (size_t k = 0; k < el.props.size(); ++k) {
                        const Prop &p = el.props[k];
                        if (p.list) {
                            size_t c = (size_t) rd.scalar(p.count_type);
                            for (size_t j = 0; j < c; ++j) rd.scalar(p.item_type);
                        } else {
                            vals[k] = rd.scalar(p.type);
                        }
                    }
                    float p[3] = {(float) vals[ix], (float) vals[iy], (float) vals[iz]};
                    const float *M = m_to_world.matrix.m;
                    for (int r = 0; r < 3; ++r) {
                        float acc = M[4 * r + 3];
                        acc = std::fmaf(M[4 * r + 0], p[0], acc);
                        acc = std::fmaf(M[4 * r + 1], p[1], acc);
                        acc = std::fmaf(M[4 * r + 2], p[2], acc);
                        m_positions[3 * i + r] = acc;
                        if (!std::isfinite(acc)) fail("mesh contains invalid vertex positions/normal data");
                    }
                    if (file_normals) {
                        float n[3] = {(float) vals[inx], (float) vals[iny], (float) vals[inz]}, q[3];
                        const float *I = m_to_world.inverse.m;
                        for (int r = 0; r < 3; ++r) {
                            float acc = I[0 + r] * n[0];
                            acc = std::fmaf(I[4 + r], n[1], acc);
                            acc = std::fmaf(I[8 + r], n[2], acc);
                            q[r] = acc;
                        }
                        float il = 1.f / std::sqrt(std::fmaf(q[2], q[2], std::fmaf(q[1], q[1], q[0] * q[0])));
                        for (int r = 0; r < 3; ++r) nrm[3 * i + r] = q[r] * il;
                    }
                    if (iu >= 0 && iv >= 0) {
                        m_texcoords[2 * i] = (float) vals[iu];
                        m_texcoords[2 * i + 1] = (float) vals[iv];
                    }
                }
            } else if (el.name == "face") {
                int il = -1;
                for (size_t k = 0; k < el.props.size(); ++k)
                    if (el.props[k].list && (el.props[k].name == "vertex_index" || el.props[k].name == "vertex_indices")) il = (int) k;
                if (il < 0) fail("vertex_index/vertex_indices property not found");
                m_faces.reserve(3 * el.count);
                for (size_t i = 0; i < el.count; ++i) {
                    for (size_t k = 0; k < el.props.size(); ++k) {
                        const Prop &p = el.props[k];
                        if (!p.list) {
                            rd.scalar(p.type);
                            continue;
                        }
                        size_t c = (size_t) rd.scalar(p.count_type);
                        if ((int) k != il) {
                            for (size_t j = 0; j < c; ++j) rd.scalar(p.item_type);
                            continue;
                        }
                        uint32_t tri[3] = {0, 0, 0};
                        for (size_t j = 0; j < c; ++j) {
                            uint32_t id = (uint32_t) rd.scalar(p.item_type);
                            if (j < 3) tri[j] = id;
                            else {
                                tri[1] = tri[2];
                                tri[2] = id;
                            }
                            if (j >= 2) m_faces.insert(m_faces.end(), tri, tri + 3);
                        }
                    }
                }
            } else {
                for (size_t i = 0; i < el.count; ++i)
                    for (const Prop &p : el.props) {
                        if (p.list) {
                            size_t c = (size_t) rd.scalar(p.count_type);
                            for (size_t j = 0; j < c; ++j) rd.scalar(p.item_type);
                        } else {
                            rd.scalar(p.type);
                        }
                    }
            }
        }
        for (uint32_t id : m_faces)
            if (id >= vertex_count()) fail("face references an invalid vertex");
        if (file_normals) m_normals.swap(nrm);
        Log(Debug, "\"%s\": read %zu faces, %zu vertices", path.c_str(), m_faces.size() / 3, (size_t) vertex_count());
        if (!face_normals && !file_normals) recompute_vertex_normals();     // ply.cpp:379-384
    }
};
BF_EXPORT_PLUGIN(PLYMesh, "Mesh", "ply", "PLY Mesh")
