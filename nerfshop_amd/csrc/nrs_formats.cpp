// nrs_formats.cpp -- on-disk formats either side of the render path, host C++, no HIP calls:
//
//   nrs_snapshot_open   Testbed::load_network_config + load_snapshot     src/testbed.cu:152-184, 3054-3087
//                       (.msgpack = nlohmann::json::to_msgpack of the network config with a "snapshot" object;
//                        .ingp = the same bytes behind zlib, zstr::ifstream)
//   nrs_edits_open      Testbed::load_edits                              src/testbed.cu:3205-3236
//                       (JSON {"edit_operators": [{"type": "cage_deformation", "proxy_cage": Cage, "interpolation_mesh": TetMesh}]},
//                        Cage / TetMesh schemas: editing/datastructures/cage.h:100-145, tet_mesh.h:136-174; Eigen vectors are
//                        arrays of 3 numbers, json_binding.h:28-57, 231-248)
//
// Both formats are read into one small value tree (MessagePack decoder / JSON parser below); nothing of nlohmann::json,
// zstr or tiny-cuda-nn is used.  The accessors hand out the arrays in the layout the rest of the C-ABI consumes.
#include <zlib.h>

#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "nrs_internal.h"

using namespace nrs;

namespace {

// what the path does not render (another encoding, light directions, ...): nrs_snapshot_open answers NRS_ERR_UNSUPPORTED, not "corrupt file"
struct Unsupported : std::runtime_error {
	// (the message quotes strings of the file: printable ASCII only, bounded -- a fuzzed file must not put raw bytes into nrs_last_error())
	static std::string clean(const std::string& m) {
		std::string o;
		for (char ch : m) { if (o.size() >= 400) break; o += (ch >= 32 && ch < 127) ? ch : '?'; }
		return o;
	}
	explicit Unsupported(const std::string& m) : std::runtime_error(clean(m)) {}
};

// ---- value tree ------------------------------------------------------------------------------------------------------
struct Value {
	enum Kind { Null, Bool, Int, Float, Str, Bin, Arr, Obj } kind = Null;
	bool b = false;
	int64_t i = 0;
	double f = 0.0;
	std::string s;                 // Str and Bin payloads
	std::vector<Value> a;
	std::vector<std::pair<std::string, Value>> o;

	const Value* find(const char* key) const {
		if (kind != Obj) return nullptr;
		for (const auto& kv : o)
			if (kv.first == key) return &kv.second;
		return nullptr;
	}
	const Value& at(const char* key) const {
		const Value* v = find(key);
		if (!v) throw std::runtime_error(std::string("missing key \"") + key + "\"");
		return *v;
	}
	bool is_number() const { return kind == Int || kind == Float; }
	double number() const {
		if (kind == Int) return (double)i;
		if (kind == Float) return f;
		if (kind == Bool) return b ? 1.0 : 0.0;
		throw std::runtime_error("expected a number");
	}
	double number_or(const char* key, double dflt) const {
		const Value* v = find(key);
		return (v && v->is_number()) ? v->number() : dflt;
	}
	std::string string_or(const char* key, const std::string& dflt) const {
		const Value* v = find(key);
		return (v && v->kind == Str) ? v->s : dflt;
	}
};

// ---- MessagePack (msgpack.org spec; what nlohmann::json::to_msgpack emits: maps with str keys, bin for binary_t) ------
struct MsgpackReader {
	const uint8_t* p;
	const uint8_t* end;
	int depth = 0;
	void need(size_t n) const {
		if ((size_t)(end - p) < n) throw std::runtime_error("msgpack: truncated input");
	}
	uint64_t be(int n) {
		need((size_t)n);
		uint64_t v = 0;
		for (int k = 0; k < n; ++k) v = (v << 8) | *p++;
		return v;
	}
	std::string bytes(size_t n) {
		need(n);
		std::string s((const char*)p, n);
		p += n;
		return s;
	}
	Value read() {
		if (++depth > 64) throw std::runtime_error("msgpack: nesting too deep");
		need(1);
		const uint8_t t = *p++;
		Value v;
		auto str = [&](size_t n) { v.kind = Value::Str; v.s = bytes(n); };
		auto bin = [&](size_t n) { v.kind = Value::Bin; v.s = bytes(n); };
		auto arr = [&](size_t n) {
			v.kind = Value::Arr;
			v.a.reserve(std::min<size_t>(n, 1u << 20));
			for (size_t k = 0; k < n; ++k) v.a.push_back(read());
		};
		auto map = [&](size_t n) {
			v.kind = Value::Obj;
			for (size_t k = 0; k < n; ++k) {
				Value key = read();
				if (key.kind != Value::Str) throw std::runtime_error("msgpack: non-string map key");
				v.o.emplace_back(std::move(key.s), read());
			}
		};
		if (t <= 0x7f) { v.kind = Value::Int; v.i = t; }
		else if (t >= 0xe0) { v.kind = Value::Int; v.i = (int8_t)t; }
		else if (t >= 0xa0 && t <= 0xbf) str(t & 0x1f);
		else if (t >= 0x90 && t <= 0x9f) arr(t & 0x0f);
		else if (t >= 0x80 && t <= 0x8f) map(t & 0x0f);
		else switch (t) {
			case 0xc0: break;
			case 0xc2: v.kind = Value::Bool; v.b = false; break;
			case 0xc3: v.kind = Value::Bool; v.b = true; break;
			case 0xc4: bin((size_t)be(1)); break;
			case 0xc5: bin((size_t)be(2)); break;
			case 0xc6: bin((size_t)be(4)); break;
			case 0xca: { uint32_t u = (uint32_t)be(4); float x; memcpy(&x, &u, 4); v.kind = Value::Float; v.f = x; break; }
			case 0xcb: { uint64_t u = be(8); double x; memcpy(&x, &u, 8); v.kind = Value::Float; v.f = x; break; }
			case 0xcc: v.kind = Value::Int; v.i = (int64_t)be(1); break;
			case 0xcd: v.kind = Value::Int; v.i = (int64_t)be(2); break;
			case 0xce: v.kind = Value::Int; v.i = (int64_t)be(4); break;
			case 0xcf: v.kind = Value::Int; v.i = (int64_t)be(8); break;
			case 0xd0: v.kind = Value::Int; v.i = (int8_t)be(1); break;
			case 0xd1: v.kind = Value::Int; v.i = (int16_t)be(2); break;
			case 0xd2: v.kind = Value::Int; v.i = (int32_t)be(4); break;
			case 0xd3: v.kind = Value::Int; v.i = (int64_t)be(8); break;
			case 0xd9: str((size_t)be(1)); break;
			case 0xda: str((size_t)be(2)); break;
			case 0xdb: str((size_t)be(4)); break;
			case 0xdc: arr((size_t)be(2)); break;
			case 0xdd: arr((size_t)be(4)); break;
			case 0xde: map((size_t)be(2)); break;
			case 0xdf: map((size_t)be(4)); break;
			// ext family (nlohmann writes binary_t with a subtype as ext): payload kept as Bin
			case 0xd4: be(1); bin(1); break;
			case 0xd5: be(1); bin(2); break;
			case 0xd6: be(1); bin(4); break;
			case 0xd7: be(1); bin(8); break;
			case 0xd8: be(1); bin(16); break;
			case 0xc7: { size_t n = (size_t)be(1); be(1); bin(n); break; }
			case 0xc8: { size_t n = (size_t)be(2); be(1); bin(n); break; }
			case 0xc9: { size_t n = (size_t)be(4); be(1); bin(n); break; }
			default: throw std::runtime_error("msgpack: reserved type byte");
		}
		--depth;
		return v;
	}
};

// ---- JSON (RFC 8259; numbers via strtod) ------------------------------------------------------------------------------
struct JsonReader {
	const char* p;
	const char* end;
	int depth = 0;
	void ws() {
		while (p < end && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) ++p;
	}
	[[noreturn]] void fail(const char* what) const { throw std::runtime_error(std::string("json: ") + what); }
	std::string str() {
		if (p >= end || *p != '"') fail("expected a string");
		++p;
		std::string s;
		while (p < end && *p != '"') {
			if (*p == '\\') {
				if (++p >= end) fail("truncated escape");
				switch (*p) {
					case 'n': s += '\n'; break;
					case 't': s += '\t'; break;
					case 'r': s += '\r'; break;
					case 'b': s += '\b'; break;
					case 'f': s += '\f'; break;
					case 'u': { // BMP code point -> UTF-8 (keys / type names in these files are ASCII)
						if (end - p < 5) fail("truncated \\u escape");
						unsigned cp = (unsigned)strtoul(std::string(p + 1, 4).c_str(), nullptr, 16);
						p += 4;
						if (cp < 0x80) s += (char)cp;
						else if (cp < 0x800) { s += (char)(0xc0 | (cp >> 6)); s += (char)(0x80 | (cp & 0x3f)); }
						else { s += (char)(0xe0 | (cp >> 12)); s += (char)(0x80 | ((cp >> 6) & 0x3f)); s += (char)(0x80 | (cp & 0x3f)); }
						break;
					}
					default: s += *p;
				}
				++p;
			} else s += *p++;
		}
		if (p >= end) fail("unterminated string");
		++p;
		return s;
	}
	Value read() {
		if (++depth > 64) fail("nesting too deep");
		ws();
		if (p >= end) fail("unexpected end");
		Value v;
		if (*p == '{') {
			v.kind = Value::Obj;
			++p; ws();
			if (p < end && *p == '}') ++p;
			else for (;;) {
				ws();
				std::string key = str();
				ws();
				if (p >= end || *p != ':') fail("expected ':'");
				++p;
				v.o.emplace_back(std::move(key), read());
				ws();
				if (p < end && *p == ',') { ++p; continue; }
				if (p < end && *p == '}') { ++p; break; }
				fail("expected ',' or '}'");
			}
		} else if (*p == '[') {
			v.kind = Value::Arr;
			++p; ws();
			if (p < end && *p == ']') ++p;
			else for (;;) {
				v.a.push_back(read());
				ws();
				if (p < end && *p == ',') { ++p; continue; }
				if (p < end && *p == ']') { ++p; break; }
				fail("expected ',' or ']'");
			}
		} else if (*p == '"') { v.kind = Value::Str; v.s = str(); }
		else if (end - p >= 4 && !strncmp(p, "true", 4)) { v.kind = Value::Bool; v.b = true; p += 4; }
		else if (end - p >= 5 && !strncmp(p, "false", 5)) { v.kind = Value::Bool; v.b = false; p += 5; }
		else if (end - p >= 4 && !strncmp(p, "null", 4)) { p += 4; }
		else {
			char* e = nullptr;
			const double d = strtod(p, &e);
			if (e == p) fail("unexpected character");
			bool integral = true;
			for (const char* q = p; q < e; ++q)
				if (*q == '.' || *q == 'e' || *q == 'E') integral = false;
			if (integral && std::fabs(d) < 9.0e18) { v.kind = Value::Int; v.i = (int64_t)strtoll(p, nullptr, 10); }
			else { v.kind = Value::Float; v.f = d; }
			p = e;
		}
		--depth;
		return v;
	}
};

std::string read_file(const char* path) {
	std::ifstream f(path, std::ios::in | std::ios::binary);
	if (!f) throw std::runtime_error(std::string("cannot open '") + path + "'");
	std::string s((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
	return s;
}
// zstr::ifstream: zlib or gzip framing is auto-detected, anything else is passed through (zstr.hpp's istreambuf)
std::string maybe_inflate(const std::string& in) {
	const bool gz = in.size() >= 2 && (uint8_t)in[0] == 0x1f && (uint8_t)in[1] == 0x8b;
	const bool zl = in.size() >= 2 && (uint8_t)in[0] == 0x78 && (((uint8_t)in[0] << 8 | (uint8_t)in[1]) % 31 == 0);
	if (!gz && !zl) return in;
	z_stream zs{};
	if (inflateInit2(&zs, 15 + 32) != Z_OK) throw std::runtime_error("zlib: inflateInit2 failed");
	std::string out;
	std::vector<char> buf(1u << 20);
	zs.next_in = (Bytef*)in.data();
	zs.avail_in = (uInt)std::min<size_t>(in.size(), 0xffffffffu);
	size_t consumed = 0;
	int rc = Z_OK;
	while (rc != Z_STREAM_END) {
		if (zs.avail_in == 0) {
			consumed = (const char*)zs.next_in - in.data();
			if (consumed >= in.size()) break;
			zs.avail_in = (uInt)std::min<size_t>(in.size() - consumed, 0xffffffffu);
		}
		zs.next_out = (Bytef*)buf.data();
		zs.avail_out = (uInt)buf.size();
		rc = inflate(&zs, Z_NO_FLUSH);
		if (rc != Z_OK && rc != Z_STREAM_END) { inflateEnd(&zs); throw std::runtime_error("zlib: corrupt stream"); }
		out.append(buf.data(), buf.size() - zs.avail_out);
		if (out.size() > (size_t)8 << 30) { inflateEnd(&zs); throw std::runtime_error("zlib: stream inflates to more than 8 GiB"); }
	}
	inflateEnd(&zs);
	if (rc != Z_STREAM_END) throw std::runtime_error("zlib: truncated stream");
	return out;
}

inline float half_to_float(uint16_t h) {
	const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1fu, man = h & 0x3ffu;
	uint32_t u;
	if (exp == 0) {
		if (man == 0) u = sign;
		else { float v = (float)man * 5.9604644775390625e-08f; memcpy(&u, &v, 4); u |= sign; }
	} else if (exp == 31) u = sign | 0x7f800000u | (man << 13);
	else u = sign | ((exp + 112u) << 23) | (man << 13);
	float f;
	memcpy(&f, &u, 4);
	return f;
}
inline uint16_t float_to_half(float f) { // round to nearest even
	uint32_t x;
	memcpy(&x, &f, 4);
	const uint32_t sign = (x >> 16) & 0x8000u, ax = x & 0x7fffffffu;
	if (ax >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | ((ax > 0x7f800000u) ? 0x200u : 0u));
	if (ax >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);
	if (ax < 0x33000001u) return (uint16_t)sign;
	const int e = (int)(ax >> 23) - 127;
	const uint32_t m = (ax & 0x7fffffu) | 0x800000u;
	const int shift = (e < -14) ? (13 + (-14 - e)) : 13;
	uint32_t kept = m >> shift;
	const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
	if (rem > half || (rem == half && (kept & 1u))) kept++;
	const uint32_t h = (e < -14) ? kept : (((uint32_t)(e + 15) << 10) + (kept - 0x400u));
	return (uint16_t)(sign | h);
}

int fmt_fail(int code, const std::string& msg) { set_last_error(msg.c_str()); return code; }

// Eigen::Vector3f list -> flat floats (json_binding.h:231-248)
// (The reference's vector writers -- json_binding.h:231-321: `for (e : vec) j.push_back(e)` -- leave the json value untouched, i.e. `null`, for an EMPTY vector, and
// its readers loop `row < j.size()`, 0 for null: null reads as an empty list here too.)
void read_vec3_list(const Value& v, std::vector<float>& out, const char* what) {
	if (v.kind == Value::Null) return;
	if (v.kind != Value::Arr) throw std::runtime_error(std::string(what) + ": expected an array");
	out.reserve(v.a.size() * 3);
	for (const Value& row : v.a) {
		if (row.kind != Value::Arr || row.a.size() != 3) throw std::runtime_error(std::string(what) + ": expected [x, y, z] rows");
		for (const Value& c : row.a) out.push_back((float)c.number());
	}
}
void read_u32_list(const Value& v, std::vector<uint32_t>& out, const char* what) {
	if (v.kind == Value::Null) return;
	if (v.kind != Value::Arr) throw std::runtime_error(std::string(what) + ": expected an array");
	out.reserve(v.a.size());
	for (const Value& c : v.a) {
		if (c.kind != Value::Int || c.i < 0 || c.i > 0xffffffffll) throw std::runtime_error(std::string(what) + ": expected unsigned integers");
		out.push_back((uint32_t)c.i);
	}
}

} // namespace

struct nrs_snapshot {
	nrs_model_desc desc{};
	uint32_t aabb_scale = 1;
	std::vector<uint16_t> params;   // fp16 bits, tcnn order
	std::vector<float> density_grid; // [5 * 128^3]
	uint32_t training_step = 0;
	bool have_camera = false;
	float camera[12] = {0};
};

struct CageOperator {
	std::string type;
	nrs_affine_duplication affine{};
	std::vector<float> vertices, original_vertices, mvc, cage_vertices, cage_original_vertices;
	std::vector<uint32_t> tets, cage_indices;
	uint32_t n_cage_vertices = 0;
};
struct nrs_edits {
	std::vector<CageOperator> ops;
};

extern "C" {

int nrs_snapshot_open(const char* path, nrs_snapshot** out) {
	if (!path || !out) return fmt_fail(NRS_ERR_INVALID_ARG, "nrs_snapshot_open: NULL argument");
	try {
		const std::string raw = maybe_inflate(read_file(path));
		MsgpackReader rd{(const uint8_t*)raw.data(), (const uint8_t*)raw.data() + raw.size()};
		const Value root = rd.read();
		const Value* snap = root.find("snapshot");
		if (!snap) throw std::runtime_error("file does not contain a snapshot");           // testbed.cu:3056
		std::unique_ptr<nrs_snapshot> s(new nrs_snapshot());
		if ((uint32_t)snap->number_or("density_grid_size", 0) != kGrid) throw std::runtime_error("Incompatible grid size in snapshot."); // :3074
		// aabb_scale: export_snapshot writes snapshot.nerf.aabb_scale (:3148); save_snapshot keeps it in the dataset (:3105, json_binding.h:154)
		const Value* nerf = snap->find("nerf");
		double aabb_scale = 1;
		if (nerf) {
			aabb_scale = nerf->number_or("aabb_scale", 0);
			if (aabb_scale <= 0) { const Value* ds = nerf->find("dataset"); aabb_scale = ds ? ds->number_or("aabb_scale", 1) : 1; }
		}
		const uint32_t as = (uint32_t)aabb_scale;
		if (as == 0 || (as & (as - 1)) || as > (1u << (kCascades - 1))) throw std::runtime_error("aabb_scale must be a power of two <= 16"); // testbed_nerf.cu:3397-3408
		s->aabb_scale = as;
		// network hyper-parameters: reset_network, testbed.cu:2257-2333
		const Value& enc = root.at("encoding");
		const Value& net = root.at("network");
		const Value* rgb = root.find("rgb_network");
		const Value* dir = root.find("dir_encoding");
		const bool has_dir = rgb && dir; // testbed.cu:2314: otherwise NerfNetworkNoDir (configs/nerf/base_nodir.json): sh_degree = 0 in nrs_model_desc
		nrs_model_desc& d = s->desc;
		// What this path renders is configs/nerf/base.json's family: HashGrid positions, SH directions, ReLU networks without output activation.  Anything else
		// tiny-cuda-nn can build (configs/nerf/{frequency,densegrid,tensor,none,...}.json; create_encoding / create_network compare otype case-insensitively) is REFUSED
		// here rather than mis-rendered.
		auto lower = [](std::string v) { for (char& ch : v) ch = (char)std::tolower((unsigned char)ch); return v; };
		const std::string enc_type = lower(enc.string_or("otype", "HashGrid"));
		if (enc_type != "hashgrid" && enc_type != "grid") throw Unsupported("encoding.otype \"" + enc.string_or("otype", "") + "\": only HashGrid is rendered by this path");
		if (lower(enc.string_or("type", "Hash")) != "hash") throw Unsupported("encoding.type \"" + enc.string_or("type", "") + "\": only the hashed grid (type Hash) is rendered by this path");
		if (lower(enc.string_or("interpolation", "Linear")) != "linear") throw Unsupported("encoding.interpolation \"" + enc.string_or("interpolation", "") + "\": only Linear is rendered by this path");
		auto check_mlp = [&](const Value& n, const char* what) {
			if (lower(n.string_or("activation", "ReLU")) != "relu") throw Unsupported(std::string(what) + ".activation \"" + n.string_or("activation", "") + "\": only ReLU");
			if (lower(n.string_or("output_activation", "None")) != "none") throw Unsupported(std::string(what) + ".output_activation \"" + n.string_or("output_activation", "") + "\": only None");
			const std::string t = lower(n.string_or("otype", "FullyFusedMLP"));
			if (t != "fullyfusedmlp" && t != "cutlassmlp" && t != "megakernelmlp") throw Unsupported(std::string(what) + ".otype \"" + n.string_or("otype", "") + "\": not a plain MLP");
		};
		check_mlp(net, "network");
		if (has_dir) {
			check_mlp(*rgb, "rgb_network");
			// configs/nerf/base.json:37-51: Composite[SphericalHarmonics on 3 dims; Identity on the rest] -- or SphericalHarmonics alone
			const std::string dt = lower(dir->string_or("otype", "OneBlob")); // (a missing otype is tiny-cuda-nn's default, OneBlob: refused below, not taken for the supported kind -- ADVICE r5)
			const Value* first = dir;
			if (dt == "composite") {
				const Value* nested = dir->find("nested");
				if (!nested || nested->kind != Value::Arr || nested->a.empty()) throw Unsupported("dir_encoding: Composite without nested encodings");
				first = &nested->a[0];
				for (size_t k = 1; k < nested->a.size(); ++k)
					if (lower(nested->a[k].string_or("otype", "OneBlob")) != "identity") throw Unsupported("dir_encoding.nested[" + std::to_string(k) + "]: only Identity may follow the spherical harmonics");
				if ((uint32_t)first->number_or("n_dims_to_encode", 3) != 3u) throw Unsupported("dir_encoding.nested[0].n_dims_to_encode must be 3");
			} else if (dt != "sphericalharmonics") throw Unsupported("dir_encoding.otype \"" + dir->string_or("otype", "") + "\": only SphericalHarmonics (alone or first in a Composite)");
			if (lower(first->string_or("otype", "OneBlob")) != "sphericalharmonics") throw Unsupported("dir_encoding: the view direction must be encoded by SphericalHarmonics");
		}
		// Light directions (NerfCoordinate::set_with_optional_light_dir, nerf.h:73-93; n_extra_dims = 3 when dataset.has_light_dirs, testbed.cu:2318): three more network
		// inputs per sample that this path does not carry.  Neither save_snapshot nor the dataset's to_json stores the flag (json_binding.h:136-160), so it is
		// recognised by the keys a writer MAY add and -- below -- by the size of the parameter blob, which such a network cannot hide.
		{
			const Value* nerf_v = snap->find("nerf");
			const Value* ds = nerf_v ? nerf_v->find("dataset") : nullptr;
			auto truthy = [](const Value* v) { return v && ((v->kind == Value::Bool && v->b) || (v->is_number() && v->number() != 0.0)); };
			if ((ds && (truthy(ds->find("has_light_dirs")) || truthy(ds->find("n_extra_dims")))) || (nerf_v && truthy(nerf_v->find("n_extra_dims"))) || truthy(root.find("n_extra_dims")))
				throw Unsupported("the snapshot was trained with light directions (has_light_dirs / n_extra_dims = 3): this path renders position + view direction only");
		}
		// every hyper-parameter is validated BEFORE it is used in arithmetic (a crafted file must be refused, not divide by zero or shift by 200)
		auto bounded = [](double v, double lo, double hi, const char* what) -> uint32_t {
			if (!(v >= lo && v <= hi) || v != std::floor(v)) throw std::runtime_error(std::string("snapshot: ") + what + " out of range");
			return (uint32_t)v;
		};
		d.n_features_per_level = bounded(enc.number_or("n_features_per_level", 2), 2, 2, "encoding.n_features_per_level (only 2 is supported)");
		const double n_features = enc.number_or("n_features", 0);
		if (!(n_features >= 0 && n_features <= 4096)) throw std::runtime_error("snapshot: encoding.n_features out of range");
		d.n_levels = n_features > 0 ? (uint32_t)n_features / d.n_features_per_level : (uint32_t)enc.number_or("n_levels", 16);
		d.log2_hashmap_size = bounded(enc.number_or("log2_hashmap_size", 15), 8, 24, "encoding.log2_hashmap_size");
		d.base_resolution = bounded(enc.number_or("base_resolution", 0), 0, 65536, "encoding.base_resolution");
		if (!d.base_resolution) d.base_resolution = 1u << (d.log2_hashmap_size / 3);
		float pls = (float)enc.number_or("per_level_scale", 0.0);
		if (pls <= 0.0f && d.n_levels > 1) pls = std::exp(std::log(2048.0f * (float)as / (float)d.base_resolution) / (float)(d.n_levels - 1));
		d.per_level_scale = pls;
		if (!std::isfinite(pls) || pls <= 0.0f) throw std::runtime_error("snapshot: encoding.per_level_scale out of range");
		d.n_neurons = bounded(net.number_or("n_neurons", 64), 1, 4096, "network.n_neurons");
		d.density_hidden_layers = bounded(net.number_or("n_hidden_layers", 1), 0, 64, "network.n_hidden_layers");
		d.density_output_dims = 16; // nerf_network_full.h:47-49
		d.rgb_hidden_layers = 0;
		d.sh_degree = 0;
		if (has_dir) {
			d.rgb_hidden_layers = bounded(rgb->number_or("n_hidden_layers", 2), 0, 64, "rgb_network.n_hidden_layers");
			if (d.rgb_hidden_layers > 0 && bounded(rgb->number_or("n_neurons", 64), 1, 4096, "rgb_network.n_neurons") != d.n_neurons) throw std::runtime_error("density / rgb networks of different widths are not supported");
			d.sh_degree = 4;
			if (const Value* nested = dir->find("nested"))
				if (nested->kind == Value::Arr && !nested->a.empty()) d.sh_degree = bounded(nested->a[0].number_or("degree", 4), 1, 16, "dir_encoding.degree");
		}
		d.rgb_activation = NRS_ACT_LOGISTIC;      // testbed.h:636-637 defaults; snapshots do not store them
		d.density_activation = NRS_ACT_EXPONENTIAL;
		const float half = 0.5f * (float)std::min<uint32_t>(1u << (kCascades - 1), as); // m_aabb, testbed_nerf.cu:3410-3411
		for (int k = 0; k < 3; ++k) { d.aabb_min[k] = 0.5f - half; d.aabb_max[k] = 0.5f + half; }
		// parameters: tcnn Trainer::serialize -> "params_binary" (+ "params_type": "__half" | "float"), "n_params"
		const Value& pb = snap->at("params_binary");
		if (pb.kind != Value::Bin) throw std::runtime_error("params_binary is not binary");
		const std::string ptype = snap->string_or("params_type", "__half");
		const size_t n_expected = nrs_model_n_params(&d);
		if (n_expected == 0) throw Unsupported("network architecture outside configs/nerf/base.json's family (hash grid 16 x 2, 64-wide density network with one hidden layer, rgb network of 0..3 hidden layers on SH degree 4 or none)");
		{ // a network with 3 extra input dimensions: the direction encoding grows from 16 to 16 + 3 -> padded to 32, i.e. the rgb network's first matrix from [64 x 32] to [64 x 48]
			const size_t n_light = n_expected + (size_t)d.n_neurons * 16u, bytes = ptype == "float" ? 4u : 2u;
			if (has_dir && d.rgb_hidden_layers > 0 && pb.s.size() == n_light * bytes)
				throw Unsupported("params_binary has the size of this architecture WITH 3 extra input dimensions (light directions, n_extra_dims = 3): not rendered by this path");
		}
		if (ptype == "float") {
			if (pb.s.size() != n_expected * 4) throw std::runtime_error("params_binary has the wrong size for this architecture");
			s->params.resize(n_expected);
			for (size_t k = 0; k < n_expected; ++k) { float f; memcpy(&f, pb.s.data() + 4 * k, 4); s->params[k] = float_to_half(f); }
		} else {
			if (pb.s.size() != n_expected * 2) throw std::runtime_error("params_binary has the wrong size for this architecture");
			s->params.resize(n_expected);
			memcpy(s->params.data(), pb.s.data(), n_expected * 2);
		}
		// density grid: float [5*128^3] from save_snapshot (:3097), fp16 [(max_cascade+1)*128^3] from export_snapshot (:3139-3146)
		const Value& gb = snap->at("density_grid_binary");
		if (gb.kind != Value::Bin) throw std::runtime_error("density_grid_binary is not binary");
		const size_t full = (size_t)kGridVol * kCascades;
		s->density_grid.assign(full, 0.f);
		uint32_t max_cascade = 0;
		while ((1u << max_cascade) < as) ++max_cascade;
		if (gb.s.size() == full * 4) memcpy(s->density_grid.data(), gb.s.data(), full * 4);
		else if (gb.s.size() == (size_t)(max_cascade + 1) * kGridVol * 2 || gb.s.size() == full * 2) {
			const size_t n = gb.s.size() / 2;
			for (size_t k = 0; k < n; ++k) { uint16_t h; memcpy(&h, gb.s.data() + 2 * k, 2); s->density_grid[k] = half_to_float(h); }
		} else throw std::runtime_error("density_grid_binary has an unexpected size");
		s->training_step = (uint32_t)snap->number_or("training_step", 0);
		if (const Value* cam = snap->find("camera"))
			if (const Value* m = cam->find("matrix"))
				if (m->kind == Value::Arr && m->a.size() == 3) { // 3 rows of 4 (Eigen to_json is row by row) -> column-major 3x4
					for (int r = 0; r < 3; ++r)
						for (int c = 0; c < 4; ++c) s->camera[3 * c + r] = (float)m->a[r].a.at(c).number();
					s->have_camera = true;
				}
		*out = s.release();
		return NRS_OK;
	} catch (const Unsupported& e) {
		return fmt_fail(NRS_ERR_UNSUPPORTED, std::string("nrs_snapshot_open('") + path + "'): unsupported: " + e.what());
	} catch (const std::exception& e) {
		return fmt_fail(NRS_ERR_INVALID_ARG, std::string("nrs_snapshot_open('") + path + "'): " + e.what());
	}
}
void nrs_snapshot_close(nrs_snapshot* s) { delete s; }
int nrs_snapshot_model_desc(const nrs_snapshot* s, nrs_model_desc* desc_out, uint32_t* aabb_scale_out) {
	if (!s || !desc_out) return fmt_fail(NRS_ERR_INVALID_ARG, "nrs_snapshot_model_desc: NULL argument");
	*desc_out = s->desc;
	if (aabb_scale_out) *aabb_scale_out = s->aabb_scale;
	return NRS_OK;
}
const void* nrs_snapshot_params_fp16(const nrs_snapshot* s, size_t* n_params_out) {
	if (!s) return nullptr;
	if (n_params_out) *n_params_out = s->params.size();
	return s->params.data();
}
const float* nrs_snapshot_density_grid(const nrs_snapshot* s, size_t* n_floats_out) {
	if (!s) return nullptr;
	if (n_floats_out) *n_floats_out = s->density_grid.size();
	return s->density_grid.data();
}
int nrs_snapshot_camera(const nrs_snapshot* s, float* camera_matrix12_out) {
	if (!s || !camera_matrix12_out) return fmt_fail(NRS_ERR_INVALID_ARG, "nrs_snapshot_camera: NULL argument");
	if (!s->have_camera) return fmt_fail(NRS_ERR_STATE, "nrs_snapshot_camera: the snapshot stores no camera");
	memcpy(camera_matrix12_out, s->camera, sizeof(s->camera));
	return NRS_OK;
}

int nrs_edits_open(const char* path, nrs_edits** out) {
	if (!path || !out) return fmt_fail(NRS_ERR_INVALID_ARG, "nrs_edits_open: NULL argument");
	try {
		const std::string raw = read_file(path);
		JsonReader rd{raw.data(), raw.data() + raw.size()};
		const Value root = rd.read();
		const Value& list = root.at("edit_operators");
		if (list.kind != Value::Arr) throw std::runtime_error("edit_operators is not an array");
		std::unique_ptr<nrs_edits> e(new nrs_edits());
		for (const Value& op : list.a) {
			CageOperator c;
			c.type = op.string_or("type", "");
			if (c.type == "cage_deformation") {
				const Value& cage = op.at("proxy_cage");                       // cage.h:100-145
				read_vec3_list(cage.at("vertices"), c.cage_vertices, "proxy_cage.vertices");
				read_vec3_list(cage.at("original_vertices"), c.cage_original_vertices, "proxy_cage.original_vertices");
				read_u32_list(cage.at("indices"), c.cage_indices, "proxy_cage.indices");
				c.n_cage_vertices = (uint32_t)(c.cage_vertices.size() / 3);
				//  nrs_edits_cage hands out both arrays under ONE count: a file whose original_vertices is shorter (or `null` beside a non-empty `vertices`) is refused
				if (c.cage_original_vertices.size() != c.cage_vertices.size()) throw std::runtime_error("proxy_cage: vertices / original_vertices differ in size");
				if (c.cage_indices.size() % 3) throw std::runtime_error("proxy_cage.indices is not a triangle list");
				for (uint32_t i : c.cage_indices)
					if (i >= c.n_cage_vertices) throw std::runtime_error("proxy_cage.indices out of range");
				if (const Value* mesh = op.find("interpolation_mesh")) {       // tet_mesh.h:136-174
					read_vec3_list(mesh->at("vertices"), c.vertices, "interpolation_mesh.vertices");
					read_vec3_list(mesh->at("original_vertices"), c.original_vertices, "interpolation_mesh.original_vertices");
					read_u32_list(mesh->at("tets"), c.tets, "interpolation_mesh.tets");
					if (c.vertices.size() != c.original_vertices.size()) throw std::runtime_error("interpolation_mesh: vertices / original_vertices differ in size");
					if (c.tets.size() % 4) throw std::runtime_error("interpolation_mesh.tets is not a multiple of 4");
					const uint32_t nv = (uint32_t)(c.vertices.size() / 3);
					for (uint32_t i : c.tets)
						if (i >= nv) throw std::runtime_error("interpolation_mesh.tets out of range");
					const Value& mvc = mesh->at("mvc_coordinates");            // std::vector<std::vector<float>>, one row per tet vertex
					if (mvc.kind != Value::Arr && mvc.kind != Value::Null) throw std::runtime_error("interpolation_mesh.mvc_coordinates is not an array");
					if (!mvc.a.empty()) {
						if (mvc.a.size() != nv) throw std::runtime_error("interpolation_mesh.mvc_coordinates: one row per vertex expected");
						c.mvc.reserve((size_t)nv * c.n_cage_vertices);
						for (const Value& row : mvc.a) {
							if (row.kind != Value::Arr || row.a.size() != c.n_cage_vertices) throw std::runtime_error("interpolation_mesh.mvc_coordinates: row length != cage vertices");
							for (const Value& w : row.a) c.mvc.push_back((float)w.number());
						}
					}
				}
			} else if (c.type == "affine_duplication") {                       // affine_duplication.h:31-40
				auto vec3 = [](const Value& v, float* out, const char* what) {
					if (v.kind != Value::Arr || v.a.size() != 3) throw std::runtime_error(std::string(what) + ": expected 3 numbers");
					for (int k = 0; k < 3; ++k) out[k] = (float)v.a[k].number();
				};
				auto mat3 = [](const Value& v, float* out, const char* what) { // rows of 3 (Eigen to_json) -> column-major
					if (v.kind != Value::Arr || v.a.size() != 3) throw std::runtime_error(std::string(what) + ": expected 3 rows");
					for (int r = 0; r < 3; ++r) {
						if (v.a[r].kind != Value::Arr || v.a[r].a.size() != 3) throw std::runtime_error(std::string(what) + ": expected 3 x 3 numbers");
						for (int col = 0; col < 3; ++col) out[3 * col + r] = (float)v.a[r].a[col].number();
					}
				};
				const Value& box = op.at("selection_box");
				vec3(box.at("center"), c.affine.selection_center, "selection_box.center");
				vec3(box.at("scale"), c.affine.selection_scale, "selection_box.scale");
				mat3(box.at("rot_matrix"), c.affine.selection_rot, "selection_box.rot_matrix");
				vec3(op.at("translation"), c.affine.translation, "translation");
				vec3(op.at("scale"), c.affine.scale, "scale");
				mat3(op.at("rotation_matrix"), c.affine.rotation, "rotation_matrix");
				c.affine.hide_original = op.at("hide_original").number() != 0.0;
				c.affine.correct_dir = op.at("correct_dir").number() != 0.0;
			} else if (c.type != "twist") {
				throw std::runtime_error("Invalid edit operator!");            // testbed.cu:3233
			}
			e->ops.push_back(std::move(c));
		}
		*out = e.release();
		return NRS_OK;
	} catch (const std::exception& ex) {
		return fmt_fail(NRS_ERR_INVALID_ARG, std::string("nrs_edits_open('") + path + "'): " + ex.what());
	}
}
void nrs_edits_close(nrs_edits* e) { delete e; }
uint32_t nrs_edits_count(const nrs_edits* e) { return e ? (uint32_t)e->ops.size() : 0; }
const char* nrs_edits_type(const nrs_edits* e, uint32_t i) { return (e && i < e->ops.size()) ? e->ops[i].type.c_str() : nullptr; }
int nrs_edits_affine(const nrs_edits* e, uint32_t i, nrs_affine_duplication* op_out) {
	if (!e || i >= e->ops.size() || !op_out) return fmt_fail(NRS_ERR_INVALID_ARG, "nrs_edits_affine: bad argument");
	if (e->ops[i].type != "affine_duplication") return fmt_fail(NRS_ERR_UNSUPPORTED, "nrs_edits_affine: operator " + std::to_string(i) + " is '" + e->ops[i].type + "'");
	*op_out = e->ops[i].affine;
	return NRS_OK;
}
int nrs_edits_cage(const nrs_edits* e, uint32_t i, nrs_tet_mesh* mesh_out, const float** h_mvc_weights_out, const float** h_cage_vertices_out,
                   const float** h_cage_original_vertices_out, const uint32_t** h_cage_triangles_out, uint32_t* n_cage_vertices_out,
                   uint32_t* n_cage_triangles_out) {
	if (!e || i >= e->ops.size() || !mesh_out) return fmt_fail(NRS_ERR_INVALID_ARG, "nrs_edits_cage: bad argument");
	const CageOperator& c = e->ops[i];
	if (c.type != "cage_deformation") return fmt_fail(NRS_ERR_UNSUPPORTED, "nrs_edits_cage: operator " + std::to_string(i) + " is '" + c.type + "'");
	// (an operator saved before its cage was tetrahedralised has no "interpolation_mesh" key, growing_selection.cu:2477: mesh_out then says n_vertices = n_tets = 0
	// with NULL arrays, and the proxy cage is handed out all the same)
	memset(mesh_out, 0, sizeof(*mesh_out));
	mesh_out->n_vertices = (uint32_t)(c.vertices.size() / 3);
	mesh_out->n_tets = (uint32_t)(c.tets.size() / 4);
	mesh_out->h_vertices = c.vertices.empty() ? nullptr : c.vertices.data();
	mesh_out->h_original_vertices = c.original_vertices.empty() ? nullptr : c.original_vertices.data();
	mesh_out->h_tets = c.tets.empty() ? nullptr : c.tets.data();
	mesh_out->residual_amplitude = 1.0f;
	mesh_out->correct_direction = 1; // GrowingSelection::m_correct_direction defaults to true (growing_selection.h)
	if (h_mvc_weights_out) *h_mvc_weights_out = c.mvc.empty() ? nullptr : c.mvc.data();
	if (h_cage_vertices_out) *h_cage_vertices_out = c.cage_vertices.empty() ? nullptr : c.cage_vertices.data();
	if (h_cage_original_vertices_out) *h_cage_original_vertices_out = c.cage_original_vertices.empty() ? nullptr : c.cage_original_vertices.data();
	if (h_cage_triangles_out) *h_cage_triangles_out = c.cage_indices.empty() ? nullptr : c.cage_indices.data();
	if (n_cage_vertices_out) *n_cage_vertices_out = c.n_cage_vertices;
	if (n_cage_triangles_out) *n_cage_triangles_out = (uint32_t)(c.cage_indices.size() / 3);
	return NRS_OK;
}

} // extern "C"
