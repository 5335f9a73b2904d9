// src/shapes/obj.cpp:72-354 — OBJMesh: v / vn / vt / f with i, i/j, i//k, i/j/k
// references, polygon fan triangulation, vertex de-duplication by index triple,
// to_world applied at load time, vertex normals recomputed when the file has none
// (unless face_normals=true).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>

#include "../render.h"
using namespace bfh;

static inline void xf_point(const Matrix4f &M, const float *p, float *o) {
    for (int i = 0; i < 3; ++i) {
        float acc = M.m[4 * i + 3];
        acc = std::fmaf(M.m[4 * i + 0], p[0], acc);
        acc = std::fmaf(M.m[4 * i + 1], p[1], acc);
        acc = std::fmaf(M.m[4 * i + 2], p[2], acc);
        o[i] = acc;
    }
}
static inline void xf_normal(const Matrix4f &Minv, const float *n, float *o) {
    // Transform * Normal uses the inverse transpose: columns of the inverse
    float r[3];
    for (int i = 0; i < 3; ++i) {
        float acc = Minv.m[0 + i] * n[0];
        acc = std::fmaf(Minv.m[4 + i], n[1], acc);
        acc = std::fmaf(Minv.m[8 + i], n[2], acc);
        r[i] = acc;
    }
    float il = 1.f / std::sqrt(std::fmaf(r[2], r[2], std::fmaf(r[1], r[1], r[0] * r[0])));
    for (int i = 0; i < 3; ++i) o[i] = r[i] * il;
}

class OBJMesh final : public Mesh {
public:
    explicit OBJMesh(const Properties &props) : Mesh(props) {
        bool face_normals = props.bool_("face_normals", false);
        bool flip_tex_coords = props.bool_("flip_tex_coords", true);
        std::string path = resolve_path(props.string("filename"));
        std::ifstream in(path);
        if (!in.good()) Throw("Error while loading OBJ file \"%s\": file not found", path.c_str());
        std::vector<float> vertices, normals, texcoords;
        struct Key {
            uint32_t v, t, n;
            bool operator<(const Key &o) const { return v != o.v ? v < o.v : (t != o.t ? t < o.t : n < o.n); }
        };
        std::map<Key, uint32_t> vertex_map;
        std::vector<Key> keys;
        std::string line;
        size_t line_no = 0;
        while (std::getline(in, line)) {
            ++line_no;
            const char *cur = line.c_str();
            while (*cur == ' ' || *cur == '\t') ++cur;
            if (cur[0] == 'v' && (cur[1] == ' ' || cur[1] == '\t')) {
                float p[3], q[3];
                char *end;
                cur += 2;
                for (int i = 0; i < 3; ++i) {
                    p[i] = strtof(cur, &end);
                    if (end == cur) Throw("Error while loading OBJ file \"%s\": could not parse line %zu", path.c_str(), line_no);
                    cur = end;
                }
                xf_point(m_to_world.matrix, p, q);
                if (!std::isfinite(q[0]) || !std::isfinite(q[1]) || !std::isfinite(q[2]))
                    Throw("Error while loading OBJ file \"%s\": mesh contains invalid vertex position data", path.c_str());
                vertices.insert(vertices.end(), q, q + 3);
            } else if (cur[0] == 'v' && cur[1] == 'n' && (cur[2] == ' ' || cur[2] == '\t')) {
                float p[3], q[3];
                char *end;
                cur += 3;
                for (int i = 0; i < 3; ++i) {
                    p[i] = strtof(cur, &end);
                    cur = end;
                }
                xf_normal(m_to_world.inverse, p, q);
                normals.insert(normals.end(), q, q + 3);
            } else if (cur[0] == 'v' && cur[1] == 't' && (cur[2] == ' ' || cur[2] == '\t')) {
                char *end;
                cur += 3;
                float u = strtof(cur, &end);
                cur = end;
                float v = strtof(cur, &end);
                if (flip_tex_coords) v = 1.f - v;
                texcoords.push_back(u);
                texcoords.push_back(v);
            } else if (cur[0] == 'f' && (cur[1] == ' ' || cur[1] == '\t')) {
                cur += 2;
                uint32_t tri[3] = {0, 0, 0};
                size_t vertex_index = 0;
                while (true) {
                    while (*cur == ' ' || *cur == '\t') ++cur;
                    if (*cur == '\0' || *cur == '\r' || *cur == '\n') break;
                    Key key{0, 0, 0};
                    uint32_t *slot[3] = {&key.v, &key.t, &key.n};
                    int type_index = 0;
                    while (true) {
                        char *end;
                        long val = strtol(cur, &end, 10);
                        if (end != cur) {
                            // negative indices are relative to the end of the respective array
                            if (val < 0) {
                                size_t cnt = type_index == 0 ? vertices.size() / 3 : (type_index == 1 ? texcoords.size() / 2 : normals.size() / 3);
                                val = (long) cnt + val + 1;
                            }
                            *slot[type_index] = (uint32_t) val;
                        }
                        cur = end;
                        if (*cur == '/' && type_index < 2) {
                            ++type_index;
                            ++cur;
                        } else {
                            break;
                        }
                    }
                    if (key.v == 0 || key.v > vertices.size() / 3)
                        Throw("Error while loading OBJ file \"%s\": reference to invalid vertex %u!", path.c_str(), key.v);
                    if (face_normals) key.n = 0;
                    uint32_t id;
                    auto it = vertex_map.find(key);
                    if (it == vertex_map.end()) {
                        id = (uint32_t) keys.size();
                        vertex_map[key] = id;
                        keys.push_back(key);
                    } else {
                        id = it->second;
                    }
                    if (vertex_index < 3) {
                        tri[vertex_index] = id;
                    } else {
                        tri[1] = tri[2];
                        tri[2] = id;
                    }
                    if (++vertex_index >= 3) m_faces.insert(m_faces.end(), tri, tri + 3);
                }
            }
        }
        m_positions.resize(3 * keys.size());
        bool use_normals = !face_normals && !normals.empty();
        if (use_normals) m_normals.assign(3 * keys.size(), 0.f);
        if (!texcoords.empty()) m_texcoords.assign(2 * keys.size(), 0.f);
        for (size_t i = 0; i < keys.size(); ++i) {
            const Key &k = keys[i];
            std::memcpy(&m_positions[3 * i], &vertices[3 * (k.v - 1)], 12);
            if (k.t) {
                if (k.t > texcoords.size() / 2) Throw("Error while loading OBJ file \"%s\": reference to invalid texture coordinate %u!", path.c_str(), k.t);
                std::memcpy(&m_texcoords[2 * i], &texcoords[2 * (k.t - 1)], 8);
            }
            if (use_normals && k.n) {
                if (k.n > normals.size() / 3) Throw("Error while loading OBJ file \"%s\": reference to invalid normal %u!", path.c_str(), k.n);
                std::memcpy(&m_normals[3 * i], &normals[3 * (k.n - 1)], 12);
            }
        }
        Log(Debug, "\"%s\": read %zu faces, %zu vertices", path.c_str(), m_faces.size() / 3, keys.size());
        if (!face_normals && normals.empty()) recompute_vertex_normals();     // obj.cpp:339-344
        if (has_vertex_texcoords())
            Log(Warn, "\"%s\": texture coordinates are ignored by the HIP path (the shading frame follows the geometric normal)", path.c_str());
    }
};
BF_EXPORT_PLUGIN(OBJMesh, "Mesh", "obj", "OBJ Mesh")
