// vx_obj.cpp -- minimal Wavefront OBJ reader for the positions + faces the voxelizer consumes.
//
// Stands where the reference calls tinyobj::ObjReader::ParseFromFile (VoxelBuilder.hpp:59-69, octTree.hpp:304-314):
// the hot path only reads attrib.vertices (xyz float32) and, per shape in file order, mesh.indices[].vertex_index in
// triples.  Shapes are flattened in order, which is the order both reference drivers walk them in.
// tinyobjloader itself is third-party and absent from the reference tree (unpinned version); this reader accepts the
// same `v` / `f` grammar (v, v/vt, v//vn, v/vt/vn, negative = relative indices, polygons fan-triangulated) and parses
// numbers with strtod -> float.  OBJ-text parity with tinyobj's own number parser / polygon triangulation is not
// pinned by anything in the reference; fixtures use `f i j k` triangles and %.9g floats only.
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace vx {

static inline const char* skip_ws(const char* p) { while (*p == ' ' || *p == '\t') ++p; return p; }

// returns 0 ok, 1 file missing/unreadable, 2 parse error (msg filled)
int load_obj(const char* path, std::vector<float>& verts, std::vector<int32_t>& tris, std::string& msg)
{
    FILE* f = std::fopen(path, "rb");
    if (!f) { msg = "Path does not exist!"; return 1; }
    std::string data;
    {
        char buf[1 << 16];
        size_t n;
        while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) data.append(buf, n);
    }
    std::fclose(f);
    verts.clear();
    tris.clear();
    std::vector<int64_t> poly;
    const char* p = data.c_str();
    const char* end = p + data.size();
    size_t line_no = 0;
    while (p < end) {
        ++line_no;
        const char* eol = (const char*)std::memchr(p, '\n', (size_t)(end - p));
        if (!eol) eol = end;
        const char* q = skip_ws(p);
        if (q[0] == 'v' && (q[1] == ' ' || q[1] == '\t')) {
            q += 2;
            float xyz[3] = {0.f, 0.f, 0.f};
            for (int k = 0; k < 3; ++k) {
                q = skip_ws(q);
                if (q >= eol) break;
                char* e = nullptr;
                const double d = std::strtod(q, &e);
                if (e == q) break;
                xyz[k] = (float)d;
                q = e;
            }
            verts.push_back(xyz[0]); verts.push_back(xyz[1]); verts.push_back(xyz[2]);
        } else if (q[0] == 'f' && (q[1] == ' ' || q[1] == '\t')) {
            q += 2;
            poly.clear();
            const int64_t nv = (int64_t)(verts.size() / 3);
            for (;;) {
                q = skip_ws(q);
                if (q >= eol || *q == '\r' || *q == '#') break;
                char* e = nullptr;
                const long long vi = std::strtoll(q, &e, 10);
                if (e == q) { msg = "malformed face at line " + std::to_string(line_no); return 2; }
                q = e;
                while (q < eol && *q != ' ' && *q != '\t' && *q != '\r') ++q;  // skip /vt/vn
                int64_t idx;
                if (vi > 0) idx = vi - 1;
                else if (vi < 0) idx = nv + vi;
                else { msg = "face index 0 at line " + std::to_string(line_no); return 2; }
                if (idx < 0 || idx >= nv) { msg = "face index out of range at line " + std::to_string(line_no); return 2; }
                poly.push_back(idx);
            }
            for (size_t k = 2; k < poly.size(); ++k) {
                tris.push_back((int32_t)poly[0]);
                tris.push_back((int32_t)poly[k - 1]);
                tris.push_back((int32_t)poly[k]);
            }
        }
        p = eol + 1;
    }
    return 0;
}

}  // namespace vx
