// vx_obj.cpp -- Wavefront OBJ (+ MTL) reader for what the voxelizer consumes: positions, faces, per-face material ids.
//
// Stands where the reference calls tinyobj::ObjReader::ParseFromFile (VoxelBuilder.hpp:59-69, octTree.hpp:304-314):
// the hot path reads attrib.vertices (xyz float32), per shape in file order mesh.indices[].vertex_index in triples, and --
// in the material plumbing the reference keeps commented out (VoxelBuilder.hpp:375-395) -- mesh.material_ids[face] and the
// material_t records (ior, dissolve, shininess, illum, ambient, diffuse, specular, transmittance, emission).  Shapes are
// flattened in order, which is the order both reference drivers walk them in.
// tinyobjloader itself is third-party and absent from the reference tree (unpinned version); this reader accepts the same
// `v` / `f` / `usemtl` / `mtllib` grammar (v, v/vt, v//vn, v/vt/vn, negative = relative indices, polygons fan-triangulated,
// positive indices may refer to vertices defined later in the file) and parses numbers locale-independently
// (std::from_chars, bounded to the line) -> float.  OBJ-text parity with tinyobj's own number parser / polygon
// triangulation is not pinned by anything in the reference; fixtures use `f i j k` triangles and %.9g floats only.
#include <cerrno>
#include <charconv>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/voxhip.h"

namespace vx {

static inline const char* skip_ws(const char* p, const char* eol) { while (p < eol && (*p == ' ' || *p == '\t')) ++p; return p; }

// one decimal number in [q, eol); never reads past the line, never consults the C locale
static bool parse_real(const char*& q, const char* eol, double& out)
{
    const char* s = q;
    if (s < eol && *s == '+') ++s;  // from_chars rejects an explicit plus sign
    const std::from_chars_result r = std::from_chars(s, eol, out);
    if (r.ec != std::errc() && r.ec != std::errc::result_out_of_range) return false;
    if (r.ptr == s) return false;
    q = r.ptr;
    return true;
}

static std::string token(const char*& q, const char* eol)
{
    q = skip_ws(q, eol);
    const char* b = q;
    while (q < eol && *q != ' ' && *q != '\t' && *q != '\r') ++q;
    return std::string(b, q);
}

static bool read_file(const char* path, std::string& data)
{
    FILE* f = std::fopen(path, "rb");
    if (!f) return false;
    char buf[1 << 16];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) data.append(buf, n);
    std::fclose(f);
    return true;
}

// tinyobj's material_t defaults (InitMaterial): everything zero except dissolve, shininess and ior = 1
static vx_material default_mtl()
{
    vx_material m;
    std::memset(&m, 0, sizeof(m));
    m.shininess = 1.0f;
    m.ior = 1.0f;
    m.dissolve = 1.0f;
    m.illum = 0;
    m.texture_id = -1;
    return m;
}

// .mtl: newmtl / Ka Kd Ks Kt|Tf Ke Ns Ni d|Tr illum.  Texture maps are not on the voxel path (textureID stays -1 as in
// the reference's copy, VoxelBuilder.hpp:383-394).
static void load_mtl(const std::string& path, std::vector<vx_material>& mats, std::unordered_map<std::string, int>& by_name)
{
    std::string data;
    if (!read_file(path.c_str(), data)) return;  // tinyobj: a missing .mtl is a warning, faces keep material id -1
    const char* p = data.c_str();
    const char* end = p + data.size();
    int cur = -1;
    while (p < end) {
        const char* eol = (const char*)std::memchr(p, '\n', (size_t)(end - p));
        if (!eol) eol = end;
        const char* q = p;
        const std::string key = token(q, eol);
        auto vec3 = [&](float* dst) {
            for (int k = 0; k < 3; ++k) {
                q = skip_ws(q, eol);
                double d;
                if (!parse_real(q, eol, d)) break;
                dst[k] = (float)d;
            }
        };
        auto real1 = [&](float& dst) {
            q = skip_ws(q, eol);
            double d;
            if (parse_real(q, eol, d)) dst = (float)d;
        };
        if (key == "newmtl") {
            const std::string name = token(q, eol);
            mats.push_back(default_mtl());
            cur = (int)mats.size() - 1;
            by_name.emplace(name, cur);  // first definition of a name wins, as in tinyobj's map insert
        } else if (cur >= 0) {
            vx_material& m = mats[(size_t)cur];
            if (key == "Ka") vec3(m.ambient);
            else if (key == "Kd") vec3(m.diffuse);
            else if (key == "Ks") vec3(m.specular);
            else if (key == "Kt" || key == "Tf") vec3(m.transmittance);
            else if (key == "Ke") vec3(m.emission);
            else if (key == "Ns") real1(m.shininess);
            else if (key == "Ni") real1(m.ior);
            else if (key == "d") real1(m.dissolve);
            else if (key == "Tr") { float tr = 0.0f; real1(tr); m.dissolve = 1.0f - tr; }
            else if (key == "illum") { float v = 0.0f; real1(v); m.illum = (int32_t)v; }
        }
        p = eol + 1;
    }
}

// returns 0 ok, 1 file missing/unreadable, 2 parse error (msg filled)
int load_obj(const char* path, std::vector<float>& verts, std::vector<int32_t>& tris, std::vector<int32_t>& tri_mat, std::vector<vx_material>& mats,
             std::string& msg)
{
    std::string data;
    if (!read_file(path, data)) { msg = "Path does not exist!"; return 1; }
    verts.clear();
    tris.clear();
    tri_mat.clear();
    mats.clear();
    std::unordered_map<std::string, int> mat_by_name;
    std::string dir(path);
    {
        const size_t slash = dir.find_last_of("/\\");
        dir = slash == std::string::npos ? std::string() : dir.substr(0, slash + 1);
    }
    std::vector<int64_t> poly;
    std::vector<size_t> tri_line;  // line of every triangle, for the range check after the last vertex is known
    int cur_mat = -1;
    const char* p = data.c_str();
    const char* end = p + data.size();
    size_t line_no = 0;
    while (p < end) {
        ++line_no;
        const char* eol = (const char*)std::memchr(p, '\n', (size_t)(end - p));
        if (!eol) eol = end;
        const char* q = skip_ws(p, eol);
        if (q + 1 < eol && q[0] == 'v' && (q[1] == ' ' || q[1] == '\t')) {
            q += 2;
            float xyz[3] = {0.f, 0.f, 0.f};
            for (int k = 0; k < 3; ++k) {
                q = skip_ws(q, eol);
                double d;
                if (!parse_real(q, eol, d)) break;
                xyz[k] = (float)d;
            }
            verts.push_back(xyz[0]); verts.push_back(xyz[1]); verts.push_back(xyz[2]);
        } else if (q + 1 < eol && q[0] == 'f' && (q[1] == ' ' || q[1] == '\t')) {
            q += 2;
            poly.clear();
            const int64_t nv = (int64_t)(verts.size() / 3);  // negative indices are relative to the vertices read so far
            for (;;) {
                q = skip_ws(q, eol);
                if (q >= eol || *q == '\r' || *q == '#') break;
                long long vi = 0;
                const char* s = q;
                if (s < eol && *s == '+') ++s;
                const std::from_chars_result r = std::from_chars(s, eol, vi);
                if (r.ec != std::errc() || r.ptr == s) { msg = "malformed face at line " + std::to_string(line_no); return 2; }
                q = r.ptr;
                while (q < eol && *q != ' ' && *q != '\t' && *q != '\r') ++q;  // skip /vt/vn
                int64_t idx;
                if (vi > 0) idx = vi - 1;  // may point at a vertex further down the file: checked at the end
                else if (vi < 0) {
                    idx = nv + vi;
                    if (idx < 0) { msg = "face index out of range at line " + std::to_string(line_no); return 2; }
                } else { msg = "face index 0 at line " + std::to_string(line_no); return 2; }
                if (idx > 0x7FFFFFFE) { msg = "face index out of range at line " + std::to_string(line_no); return 2; }
                poly.push_back(idx);
            }
            for (size_t k = 2; k < poly.size(); ++k) {
                tris.push_back((int32_t)poly[0]);
                tris.push_back((int32_t)poly[k - 1]);
                tris.push_back((int32_t)poly[k]);
                tri_mat.push_back(cur_mat);
                tri_line.push_back(line_no);
            }
        } else if (eol - q > 7 && std::strncmp(q, "usemtl", 6) == 0 && (q[6] == ' ' || q[6] == '\t')) {
            q += 7;
            const std::string name = token(q, eol);
            const auto it = mat_by_name.find(name);
            cur_mat = it == mat_by_name.end() ? -1 : it->second;  // unknown name: tinyobj warns and leaves the faces without material
        } else if (eol - q > 7 && std::strncmp(q, "mtllib", 6) == 0 && (q[6] == ' ' || q[6] == '\t')) {
            q += 7;
            for (;;) {
                const std::string name = token(q, eol);
                if (name.empty()) break;
                load_mtl(dir + name, mats, mat_by_name);
            }
        }
        p = eol + 1;
    }
    const int64_t nv = (int64_t)(verts.size() / 3);
    for (size_t t = 0; t < tris.size(); ++t)
        if (tris[t] >= nv) { msg = "face index out of range at line " + std::to_string(tri_line[t / 3]); return 2; }
    return 0;
}

}  // namespace vx
