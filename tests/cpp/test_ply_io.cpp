// pcp::io::read_ply / write_ply (include/pcp/io/ply.hpp) on the CPU: the format facts of SURVEY.md section 8f-1.
// usage: test_ply_io <stanford_bunny.ply> <tmp dir>
#include <pcp/io/ply.hpp>
#include <pcp/pcp.hpp>

#include <cmath>
#include <cstdio>
#include <filesystem>
#include <fstream>
#include <sstream>

static int g_failures = 0;
#define REQUIRE(cond)                                                              \
    do {                                                                           \
        if (!(cond)) {                                                             \
            std::printf("REQUIRE failed at %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            ++g_failures;                                                          \
        }                                                                          \
    } while (0)

using pcp::normal_t;
using pcp::point_t;
namespace io = pcp::io;

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    std::filesystem::path const bunny{argv[1]}, tmp{argv[2]};

    // the reference's data file: binary little endian, "obj_info" and comment lines, xyz only, 35 947 vertices
    auto [points, normals] = io::read_ply<point_t, normal_t>(bunny);
    REQUIRE(points.size() == 35947u);
    REQUIRE(normals.empty());
    float lo = 1e30f, hi = -1e30f;
    for (auto const& p : points)
    {
        REQUIRE(std::isfinite(p.x()) && std::isfinite(p.y()) && std::isfinite(p.z()));
        lo = std::min(lo, p.x());
        hi = std::max(hi, p.x());
    }
    REQUIRE(lo < hi);
    std::printf("bunny: %zu points, first (%.9g %.9g %.9g)\n", points.size(), points[0].x(), points[0].y(), points[0].z());

    // round trips in the three formats, with a separate normal block
    std::vector<point_t> p{{0.f, 1.f, 2.f}, {-1.5f, 2.25f, 1e-3f}, {3.f, -4.f, 5.f}};
    std::vector<normal_t> n{{0.f, 0.f, 1.f}, {0.6f, 0.f, 0.8f}};
    for (auto fmt : {io::ply_format_t::ascii, io::ply_format_t::binary_little_endian, io::ply_format_t::binary_big_endian})
    {
        auto const f = tmp / ("rt" + std::to_string(static_cast<int>(fmt)) + ".ply");
        io::write_ply(f, p, n, fmt);
        auto [rp, rn] = io::read_ply<point_t, normal_t>(f);
        REQUIRE(rp.size() == p.size() && rn.size() == n.size());
        for (std::size_t i = 0; i < rp.size() && i < p.size(); ++i)
            REQUIRE(std::abs(rp[i].x() - p[i].x()) < 1e-6f && std::abs(rp[i].y() - p[i].y()) < 1e-6f && std::abs(rp[i].z() - p[i].z()) < 1e-6f);
        for (std::size_t i = 0; i < rn.size() && i < n.size(); ++i) REQUIRE(rn[i] == n[i]);
        if (fmt != io::ply_format_t::ascii)
            for (std::size_t i = 0; i < rp.size() && i < p.size(); ++i) REQUIRE(rp[i].x() == p[i].x() && rp[i].z() == p[i].z());
    }
    {   // the header always carries both element blocks, ascii components are printed by std::to_string
        std::ostringstream os;
        io::write_ply<point_t, normal_t>(os, p, {}, io::ply_format_t::ascii);
        std::string const s = os.str();
        REQUIRE(s.find("element vertex 3\nproperty float x\nproperty float y\nproperty float z\n") != std::string::npos);
        REQUIRE(s.find("element normal 0\nproperty float nx\nproperty float ny\nproperty float nz\nend_header\n") != std::string::npos);
        REQUIRE(s.find("0.000000 1.000000 2.000000\n") != std::string::npos);
    }
    {   // binary records are 3 x 4 bytes even when the header says double (reference ply.hpp:741-764)
        auto const f = tmp / "declared_double.ply";
        std::ofstream o(f, std::ios::binary);
        o << "ply\nformat binary_little_endian 1.0\ncomment x\nelement vertex 2\nproperty double x\nproperty double y\nproperty double z\nend_header\n";
        float const raw[6] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f};
        o.write(reinterpret_cast<char const*>(raw), sizeof raw);
        o.close();
        auto [rp, rn] = io::read_ply<point_t, normal_t>(f);
        REQUIRE(rp.size() == 2u && rp[1].y() == 5.f && rn.empty());
    }
    {   // failures give empty results: missing file, wrong extension, not a ply, misnamed or list properties, short body
        auto [a, b] = io::read_ply<point_t, normal_t>(tmp / "missing.ply");
        REQUIRE(a.empty() && b.empty());
        auto const txt = tmp / "cloud.txt";
        { std::ofstream o(txt); o << "ply\nformat ascii 1.0\nelement vertex 0\nproperty float x\nproperty float y\nproperty float z\nend_header\n"; }
        REQUIRE(std::get<0>(io::read_ply<point_t, normal_t>(txt)).empty());
        auto bad = [&](char const* name, char const* text) {
            auto const f = tmp / name;
            { std::ofstream o(f, std::ios::binary); o << text; }
            auto [x, y] = io::read_ply<point_t, normal_t>(f);
            return x.empty() && y.empty();
        };
        REQUIRE(bad("a.ply", "plx\nformat ascii 1.0\nend_header\n"));
        REQUIRE(bad("b.ply", "ply\nformat ascii 1.0\nelement vertex 1\nproperty float x\nproperty float z\nproperty float y\nend_header\n1 2 3\n"));
        REQUIRE(bad("c.ply", "ply\nformat ascii 1.0\nelement vertex 1\nproperty list uchar int x\nproperty float y\nproperty float z\nend_header\n1 2 3\n"));
        REQUIRE(bad("d.ply", "ply\nformat ascii 1.0\nelement vertex 1\nproperty float x\nproperty double y\nproperty float z\nend_header\n1 2 3\n"));
        REQUIRE(bad("e.ply", "ply\nformat binary_little_endian 1.0\nelement vertex 4\nproperty float x\nproperty float y\nproperty float z\nend_header\nshort"));
        REQUIRE(bad("f.ply", "ply\nformat ascii 1.0\nelement vertex 2\nproperty float x\nproperty float y\nproperty float z\nend_header\n1 2 3\n"));
    }
    {   // ascii with extra columns and other elements in the header: the first three tokens count, the rest is ignored
        auto const f = tmp / "extra.ply";
        { std::ofstream o(f); o << "ply\nformat ascii 1.0\nobj_info made up\nelement vertex 2\nproperty float x\nproperty float y\nproperty float z\nelement face 0\nproperty list uchar int vertex_indices\nend_header\n1 2 3 9 9\n4 5 6\n"; }
        auto [rp, rn] = io::read_ply<point_t, normal_t>(f);
        REQUIRE(rp.size() == 2u && rp[0].z() == 3.f && rp[1].x() == 4.f);
    }
    if (g_failures == 0) std::printf("ply io: all checks passed\n");
    return g_failures == 0 ? 0 : 1;
}
