// pcp::io::read_ply / pcp::io::write_ply -- drop-in for the point-cloud subset of include/pcp/io/ply.hpp of the
// reference: the input format of the hot path ("same PLY input").
//
// Format facts reproduced (reference file:line):
//   * read_ply(path): the path must name an existing file with extension ".ply" (ply.hpp:107-126);
//   * header (ply.hpp:141-265): first line "ply"; "comment" lines skipped; "format <ascii|binary_little_endian|
//     binary_big_endian> <version>" (unknown format strings mean ascii, ply.hpp:63-73); "element vertex N" must be
//     followed by exactly three "property <type> x|y|z" lines of one type (a "list" property is an error); normals are a
//     SEPARATE "element normal M" block with nx, ny, nz (ply.hpp:239-250) -- not per-vertex properties; every other
//     header line (obj_info, other elements and their properties) is ignored; "end_header" ends it;
//   * binary bodies hold N then M records of 3 x 4-byte floats whatever type the header declared (ply.hpp:741-764),
//     byte-swapped when the file's endianness is not the machine's (ply.hpp:806-833);
//   * ascii bodies hold one record per line, the first three tokens are the components (ply.hpp:689-716);
//   * any failure gives an empty tuple, nothing throws (ply.hpp:111-123, :207, :234);
//   * write_ply always emits both element blocks ("element normal 0" for a cloud without normals), ascii components
//     as std::to_string prints them (ply.hpp:311-460).
// Mesh (face) overloads and OBJ are outside the hot path's scope.
//
// Written for this repository: the header is parsed into a small description first, bodies are read with ONE bulk
// read per element block instead of one stream read per record.
#ifndef PCP_IO_PLY_HPP
#define PCP_IO_PLY_HPP

#include "pcp/common/normals/normal.hpp"
#include "pcp/common/points/point.hpp"

#include <array>
#include <cstdint>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <istream>
#include <ostream>
#include <sstream>
#include <string>
#include <tuple>
#include <type_traits>
#include <vector>

namespace pcp {
namespace io {

enum class ply_format_t { ascii, binary_little_endian, binary_big_endian };
enum class ply_coordinate_type_t { single_precision, double_precision };

struct ply_parameters_t
{
    ply_format_t format                         = ply_format_t::ascii;
    std::size_t vertex_count                    = 0u;
    std::size_t normal_count                    = 0u;
    ply_coordinate_type_t vertex_component_type = ply_coordinate_type_t::single_precision;
    ply_coordinate_type_t normal_component_type = ply_coordinate_type_t::single_precision;
};

inline ply_format_t string_to_format(std::string const& s)
{
    if (s == "binary_little_endian") return ply_format_t::binary_little_endian;
    if (s == "binary_big_endian") return ply_format_t::binary_big_endian;
    return ply_format_t::ascii;
}

inline bool is_machine_little_endian()
{
    std::uint32_t const one = 1u;
    unsigned char first;
    std::memcpy(&first, &one, 1);
    return first == 1u;
}
inline bool is_machine_big_endian() { return !is_machine_little_endian(); }

namespace detail {

inline std::vector<std::string> words(std::string const& line)
{
    std::vector<std::string> out;
    std::istringstream ss(line);
    for (std::string w; ss >> w;) out.push_back(w);
    return out;
}

// the three "property <type> <name>" lines that must follow an element line
inline bool three_components(std::istream& is, char const* const (&names)[3], ply_coordinate_type_t& type)
{
    ply_coordinate_type_t seen[3];
    for (int i = 0; i < 3; ++i)
    {
        std::string line;
        if (!std::getline(is, line)) return false;
        auto const w = words(line);
        if (w.size() < 3 || w[0] != "property" || w[1] == "list") return false;
        seen[i] = w[1] == "double" ? ply_coordinate_type_t::double_precision : ply_coordinate_type_t::single_precision;
        if (w.back() != names[i]) return false;
    }
    type = seen[2];
    return seen[0] == seen[2] && seen[1] == seen[2];
}

inline bool parse_header(std::istream& is, ply_parameters_t& p)
{
    static char const* const xyz[3]    = {"x", "y", "z"};
    static char const* const nxnynz[3] = {"nx", "ny", "nz"};
    std::string line;
    if (!std::getline(is, line)) return false;
    {
        auto const w = words(line);
        if (w.empty() || w[0] != "ply") return false;
    }
    while (std::getline(is, line))
    {
        auto const w = words(line);
        if (w.empty() || w[0] == "comment") continue;
        if (w[0] == "end_header") return true;
        try
        {
            if (w[0] == "format" && w.size() >= 2) p.format = string_to_format(w[1]);
            else if (w[0] == "element" && w.size() >= 3 && w[1] == "vertex")
            {
                p.vertex_count = static_cast<std::size_t>(std::stoull(w.back()));
                if (!three_components(is, xyz, p.vertex_component_type)) return false;
            }
            else if (w[0] == "element" && w.size() >= 3 && w[1] == "normal")
            {
                p.normal_count = static_cast<std::size_t>(std::stoull(w.back()));
                if (!three_components(is, nxnynz, p.normal_component_type)) return false;
            }
        }
        catch (...)
        {
            return false;  // a count that is not a number
        }
    }
    return true;  // no end_header: the body is empty and the block reads below will fail or give nothing
}

// `count` records of 3 x 4-byte floats, one bulk read; swap = file endianness differs from the machine's
inline bool read_float_triples(std::istream& is, std::size_t count, bool swap, std::vector<float>& out)
{
    out.resize(count * 3u);
    if (count == 0u) return true;
    std::vector<unsigned char> raw(count * 12u);
    is.read(reinterpret_cast<char*>(raw.data()), static_cast<std::streamsize>(raw.size()));
    if (is.bad() || static_cast<std::size_t>(is.gcount()) != raw.size()) return false;
    if (swap)
        for (std::size_t i = 0; i + 3 < raw.size(); i += 4)
        {
            std::swap(raw[i], raw[i + 3]);
            std::swap(raw[i + 1], raw[i + 2]);
        }
    std::memcpy(out.data(), raw.data(), raw.size());
    return true;
}

inline bool read_ascii_triples(std::istream& is, std::size_t count, std::vector<float>& out)
{
    out.clear();
    out.reserve(count * 3u);
    std::string line;
    for (std::size_t i = 0; i < count; ++i)
    {
        if (!std::getline(is, line)) return false;
        auto const w = words(line);
        if (w.size() < 3) return false;
        try
        {
            for (int c = 0; c < 3; ++c) out.push_back(std::stof(w[static_cast<std::size_t>(c)]));
        }
        catch (...)
        {
            return false;
        }
    }
    return true;
}

inline void append_floats(std::string& bytes, float a, float b, float c, bool swap)
{
    float const v[3] = {a, b, c};
    for (float f : v)
    {
        unsigned char r[4];
        std::memcpy(r, &f, 4);
        if (swap)
        {
            std::swap(r[0], r[3]);
            std::swap(r[1], r[2]);
        }
        bytes.append(reinterpret_cast<char const*>(r), 4);
    }
}

} // namespace detail

template <class Point, class Normal>
inline auto read_ply(std::istream& is) -> std::tuple<std::vector<Point>, std::vector<Normal>>
{
    ply_parameters_t params;
    if (!detail::parse_header(is, params)) return {};
    std::vector<float> v, n;
    if (params.format == ply_format_t::ascii)
    {
        if (!detail::read_ascii_triples(is, params.vertex_count, v) || !detail::read_ascii_triples(is, params.normal_count, n))
            return {};
    }
    else
    {
        bool const file_is_little = params.format == ply_format_t::binary_little_endian;
        bool const swap           = file_is_little != is_machine_little_endian();
        if (!detail::read_float_triples(is, params.vertex_count, swap, v) ||
            !detail::read_float_triples(is, params.normal_count, swap, n))
            return {};
    }
    std::vector<Point> points;
    std::vector<Normal> normals;
    points.reserve(params.vertex_count);
    normals.reserve(params.normal_count);
    using PT = typename Point::coordinate_type;
    using NT = typename Normal::component_type;
    for (std::size_t i = 0; i < params.vertex_count; ++i)
        points.push_back(Point{static_cast<PT>(v[3 * i]), static_cast<PT>(v[3 * i + 1]), static_cast<PT>(v[3 * i + 2])});
    for (std::size_t i = 0; i < params.normal_count; ++i)
    {
        // component-wise: a normal type may normalise in its 3-argument constructor, the file's values are kept as read
        Normal nn;
        nn.nx(static_cast<NT>(n[3 * i]));
        nn.ny(static_cast<NT>(n[3 * i + 1]));
        nn.nz(static_cast<NT>(n[3 * i + 2]));
        normals.push_back(nn);
    }
    return std::make_tuple(std::move(points), std::move(normals));
}

template <class Point, class Normal>
inline auto read_ply(std::filesystem::path const& path) -> std::tuple<std::vector<Point>, std::vector<Normal>>
{
    std::error_code ec;
    if (!path.has_filename() || !path.has_extension() || path.extension() != ".ply") return {};
    if (!std::filesystem::exists(path, ec) || ec) return {};
    std::ifstream fs{path.string(), std::ios::binary};
    if (!fs.is_open()) return {};
    return read_ply<Point, Normal>(fs);
}

template <class Point, class Normal>
inline void write_ply(std::ostream& os, std::vector<Point> const& vertices, std::vector<Normal> const& normals,
                      ply_format_t format = ply_format_t::ascii)
{
    char const* const vt = std::is_same_v<typename Point::coordinate_type, double> ? "double" : "float";
    char const* const nt = std::is_same_v<typename Normal::component_type, double> ? "double" : "float";
    char const* const fmt = format == ply_format_t::ascii                  ? "ascii"
                            : format == ply_format_t::binary_little_endian ? "binary_little_endian"
                                                                           : "binary_big_endian";
    os << "ply\nformat " << fmt << " 1.0\n"
       << "element vertex " << vertices.size() << "\nproperty " << vt << " x\nproperty " << vt << " y\nproperty " << vt << " z\n"
       << "element normal " << normals.size() << "\nproperty " << nt << " nx\nproperty " << nt << " ny\nproperty " << nt
       << " nz\nend_header\n";
    if (format == ply_format_t::ascii)
    {
        for (auto const& p : vertices) os << std::to_string(p.x()) << " " << std::to_string(p.y()) << " " << std::to_string(p.z()) << "\n";
        for (auto const& n : normals) os << std::to_string(n.nx()) << " " << std::to_string(n.ny()) << " " << std::to_string(n.nz()) << "\n";
        return;
    }
    bool const swap = (format == ply_format_t::binary_little_endian) != is_machine_little_endian();
    std::string bytes;
    bytes.reserve((vertices.size() + normals.size()) * 12u);
    for (auto const& p : vertices)
        detail::append_floats(bytes, static_cast<float>(p.x()), static_cast<float>(p.y()), static_cast<float>(p.z()), swap);
    for (auto const& n : normals)
        detail::append_floats(bytes, static_cast<float>(n.nx()), static_cast<float>(n.ny()), static_cast<float>(n.nz()), swap);
    os.write(bytes.data(), static_cast<std::streamsize>(bytes.size()));
}

template <class Point, class Normal>
inline void write_ply(std::filesystem::path const& path, std::vector<Point> const& vertices, std::vector<Normal> const& normals,
                      ply_format_t format = ply_format_t::ascii)
{
    if (!path.has_filename() || !path.has_extension() || path.extension() != ".ply") return;
    if (vertices.empty()) return;  // (reference ply.hpp:289-290)
    std::ofstream ofs{path.string(), std::ios::binary};
    if (!ofs.is_open()) return;
    write_ply<Point, Normal>(ofs, vertices, normals, format);
}

} // namespace io
} // namespace pcp

#endif
