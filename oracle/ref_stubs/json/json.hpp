// Stand-in for <json/json.hpp> (nlohmann/json, vendored by the reference through tiny-cuda-nn's dependencies: absent here, version un-pinned).
// TEST INFRASTRUCTURE: used only to build oracle/_ref/libref_json.so from the reference's OWN serialisation code (json_binding.h, mesh.h, tet_mesh.h,
// cage.h, affine_bounding_box.cuh and the to_json members cut out of the .cu files), so that files written by that code pin nrs_edits_open.
//
// What is modelled, because the reference's writers depend on it:
//   * value kinds null / object / array / string / boolean / integer / unsigned / float (floats are stored as double, as nlohmann's number_float_t);
//   * objects keep their keys SORTED (nlohmann::json's object_t is a std::map);
//   * `json j = value` / `j[key] = value` / `j.push_back(value)` convert through an unqualified, argument-dependent call of to_json(j, value) made from inside
//     nlohmann::detail (nlohmann's to_json_fn): the built-in conversions (arithmetic, bool, strings, std::vector, std::map) live in nlohmann::detail and
//     user conversions are found in the namespaces associated with the VALUE's type -- so `j["x"] = std::vector<Eigen::Vector3f>` takes the built-in array
//     conversion (elements through Eigen::to_json), not ngp::to_json(std::vector<Eigen::Vector3f>), exactly as with the real library; explicit calls
//     `to_json(j["x"], v)` inside namespace ngp resolve by plain C++ rules in the reference's own code;
//   * operator[] on null creates an object (string key) or array (index); push_back / emplace_back on null create an array;
//   * dump(): integers as integers, floats with 17 significant digits (round-trip exact; nlohmann prints the SHORTEST round-trip form -- the text differs, the
//     parsed doubles are identical), no whitespace, keys in sorted order.  parse(): RFC 8259.
//   * binary values (json::binary) and to_msgpack() as nlohmann 3.8+ writes it -- for Testbed::save_snapshot's container (oracle/ref_json.cpp).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <istream>
#include <iterator>
#include <map>
#include <ostream>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

namespace nlohmann {

class json;
namespace detail {
struct to_json_fn { template <class T> void operator()(json& j, T&& v) const; };
struct from_json_fn { template <class T> void operator()(const json& j, T& v) const; };
} // namespace detail

template <class T> struct is_std_vector : std::false_type {};
template <class T, class A> struct is_std_vector<std::vector<T, A>> : std::true_type {};
namespace detail {
// is there a user from_json(const json&, T&) reachable by argument-dependent lookup (T's own namespaces)?  Standard-library types have none.
template <class T, class = void> struct has_adl_from_json_impl : std::false_type {};
template <class T> struct has_adl_from_json_impl<T, decltype(from_json(std::declval<const json&>(), std::declval<T&>()), void())> : std::true_type {};
} // namespace detail
template <class T> struct has_adl_from_json : detail::has_adl_from_json_impl<T> {};

class json {
public:
	enum class value_t : uint8_t { null, object, array, string, boolean, number_integer, number_unsigned, number_float, binary };
	typedef std::vector<json> array_t;
	typedef std::vector<std::pair<std::string, json>> object_t; // kept sorted by key
	typedef std::size_t size_type;

	json() = default;
	json(std::nullptr_t) {}
	json(const json&) = default;
	json(json&&) = default;
	json& operator=(const json&) = default;
	json& operator=(json&&) = default;
	template <class T, typename = typename std::enable_if<!std::is_same<typename std::decay<T>::type, json>::value>::type>
	json(T&& v) { detail::to_json_fn{}(*this, std::forward<T>(v)); }

	static json array() { json j; j.t = value_t::array; return j; }
	static json binary(std::vector<uint8_t> bytes) { json j; j.t = value_t::binary; j.s.assign(bytes.begin(), bytes.end()); return j; } // (nlohmann::json::binary: a byte container)
	bool is_binary() const { return t == value_t::binary; }
	const std::string& raw_binary() const { return s; }
	static json object() { json j; j.t = value_t::object; return j; }

	value_t type() const { return t; }
	bool is_null() const { return t == value_t::null; }
	bool is_object() const { return t == value_t::object; }
	bool is_array() const { return t == value_t::array; }
	bool is_string() const { return t == value_t::string; }
	bool is_boolean() const { return t == value_t::boolean; }
	bool is_number() const { return t == value_t::number_integer || t == value_t::number_unsigned || t == value_t::number_float; }
	bool is_number_float() const { return t == value_t::number_float; }
	bool is_number_integer() const { return t == value_t::number_integer || t == value_t::number_unsigned; }

	// ---- element access ----
	json& operator[](const std::string& key) {
		if (t == value_t::null) t = value_t::object;
		if (t != value_t::object) throw std::runtime_error("json: operator[] with a string key on a non-object");
		auto it = std::lower_bound(o.begin(), o.end(), key, [](const std::pair<std::string, json>& kv, const std::string& k) { return kv.first < k; });
		if (it == o.end() || it->first != key) it = o.insert(it, std::make_pair(key, json()));
		return it->second;
	}
	const json& operator[](const std::string& key) const { return at(key); }
	template <class C> json& operator[](C* key) { return (*this)[std::string(key)]; }
	template <class C> const json& operator[](C* key) const { return at(std::string(key)); }
	json& operator[](size_type i) {
		if (t == value_t::null) t = value_t::array;
		if (t != value_t::array) throw std::runtime_error("json: operator[] with an index on a non-array");
		if (i >= a.size()) a.resize(i + 1);
		return a[i];
	}
	const json& operator[](size_type i) const { return at(i); }
	json& at(const std::string& key) { return const_cast<json&>(static_cast<const json*>(this)->at(key)); }
	const json& at(const std::string& key) const {
		if (t != value_t::object) throw std::out_of_range("json: at(\"" + key + "\") on a non-object");
		auto it = std::lower_bound(o.begin(), o.end(), key, [](const std::pair<std::string, json>& kv, const std::string& k) { return kv.first < k; });
		if (it == o.end() || it->first != key) throw std::out_of_range("json: key \"" + key + "\" not found");
		return it->second;
	}
	template <class C> json& at(C* key) { return at(std::string(key)); }
	template <class C> const json& at(C* key) const { return at(std::string(key)); }
	json& at(size_type i) { return const_cast<json&>(static_cast<const json*>(this)->at(i)); }
	const json& at(size_type i) const {
		if (t != value_t::array || i >= a.size()) throw std::out_of_range("json: array index out of range");
		return a[i];
	}
	bool contains(const std::string& key) const {
		if (t != value_t::object) return false;
		auto it = std::lower_bound(o.begin(), o.end(), key, [](const std::pair<std::string, json>& kv, const std::string& k) { return kv.first < k; });
		return it != o.end() && it->first == key;
	}
	size_type size() const { return t == value_t::array ? a.size() : (t == value_t::object ? o.size() : (t == value_t::null ? 0 : 1)); }
	bool empty() const { return size() == 0; }

	void push_back(json v) {
		if (t == value_t::null) t = value_t::array;
		if (t != value_t::array) throw std::runtime_error("json: push_back on a non-array");
		a.push_back(std::move(v));
	}
	template <class... Args> json& emplace_back(Args&&... args) {
		if (t == value_t::null) t = value_t::array;
		if (t != value_t::array) throw std::runtime_error("json: emplace_back on a non-array");
		a.emplace_back(std::forward<Args>(args)...);
		return a.back();
	}
	array_t::iterator begin() { return a.begin(); }
	array_t::iterator end() { return a.end(); }
	array_t::const_iterator begin() const { return a.begin(); }
	array_t::const_iterator end() const { return a.end(); }
	const object_t& items_sorted() const { return o; }

	// ---- conversions out ----
	template <class T> T get() const { T v{}; detail::from_json_fn{}(*this, v); return v; }
	// (as nlohmann constrains it: only to types a from_json exists for, so that `std::string(j)` sees ONE viable constructor)
	template <class T, typename = typename std::enable_if<std::is_arithmetic<T>::value || std::is_enum<T>::value || std::is_same<T, std::string>::value ||
	                                                       is_std_vector<T>::value || (std::is_class<T>::value && !std::is_same<T, json>::value && has_adl_from_json<T>::value)>::type>
	operator T() const { return get<T>(); }
	template <class T> T value(const std::string& key, const T& dflt) const { return contains(key) ? at(key).template get<T>() : dflt; }
	std::string value(const std::string& key, const char* dflt) const { return contains(key) ? at(key).get<std::string>() : std::string(dflt); }

	// raw accessors for the built-in conversions
	bool raw_bool() const { return b; }
	int64_t raw_int() const { return i; }
	uint64_t raw_uint() const { return u; }
	double raw_float() const { return f; }
	const std::string& raw_string() const { return s; }
	void set_bool(bool v) { *this = json(); t = value_t::boolean; b = v; }
	void set_int(int64_t v) { *this = json(); t = value_t::number_integer; i = v; }
	void set_uint(uint64_t v) { *this = json(); t = value_t::number_unsigned; u = v; }
	void set_float(double v) { *this = json(); t = value_t::number_float; f = v; }
	void set_string(std::string v) { *this = json(); t = value_t::string; s = std::move(v); }
	void set_array() { *this = json(); t = value_t::array; }
	void set_object() { *this = json(); t = value_t::object; }
	double number() const {
		switch (t) {
		case value_t::number_integer: return (double)i;
		case value_t::number_unsigned: return (double)u;
		case value_t::number_float: return f;
		case value_t::boolean: return b ? 1.0 : 0.0;
		default: throw std::runtime_error("json: type must be number");
		}
	}

	// ---- text ----
	std::string dump(int /*indent*/ = -1) const { std::string out; dump_to(out); return out; }
	// parse(stream, callback, allow_exceptions, ignore_comments) as load_network_config calls it (testbed.cu:93, :179): // and /* */ comments are skipped
	static json parse(std::istream& is, std::nullptr_t = nullptr, bool /*allow_exceptions*/ = true, bool ignore_comments = false) {
		std::string text((std::istreambuf_iterator<char>(is)), std::istreambuf_iterator<char>());
		if (ignore_comments) {
			std::string out;
			bool in_string = false;
			for (size_t k = 0; k < text.size(); ++k) {
				const char c = text[k];
				if (in_string) { out += c; if (c == '\\' && k + 1 < text.size()) out += text[++k]; else if (c == '"') in_string = false; continue; }
				if (c == '"') { in_string = true; out += c; continue; }
				if (c == '/' && k + 1 < text.size() && text[k + 1] == '/') { while (k < text.size() && text[k] != '\n') ++k; out += '\n'; continue; }
				if (c == '/' && k + 1 < text.size() && text[k + 1] == '*') { k += 2; while (k + 1 < text.size() && !(text[k] == '*' && text[k + 1] == '/')) ++k; ++k; out += ' '; continue; }
				out += c;
			}
			text.swap(out);
		}
		return parse(text);
	}
	// RFC 7386, as nlohmann::json::merge_patch: objects merge key by key, a null value removes the key, anything else replaces
	void merge_patch(const json& patch) {
		if (!patch.is_object()) { *this = patch; return; }
		if (!is_object()) *this = object();
		for (const auto& kv : patch.o) {
			if (kv.second.is_null()) {
				auto it = std::lower_bound(o.begin(), o.end(), kv.first, [](const std::pair<std::string, json>& e, const std::string& k) { return e.first < k; });
				if (it != o.end() && it->first == kv.first) o.erase(it);
			} else {
				(*this)[kv.first].merge_patch(kv.second);
			}
		}
	}
	static json parse(const std::string& text) {
		size_t p = 0;
		json j = parse_value(text, p, 0);
		skip_ws(text, p);
		if (p != text.size()) throw std::runtime_error("json: trailing characters");
		return j;
	}
	// nlohmann::json::to_msgpack (binary_writer::write_msgpack, 3.8+): the smallest integer encoding, float32 when the double is exactly a float (else float64),
	// maps with their keys in sorted order, bin 8 / 16 / 32 for binary values.
	static void to_msgpack(const json& j, std::ostream& os) { std::string out; j.msgpack_to(out); os.write(out.data(), (std::streamsize)out.size()); }
	static std::vector<uint8_t> to_msgpack(const json& j) { std::string out; j.msgpack_to(out); return std::vector<uint8_t>(out.begin(), out.end()); }
	friend std::ostream& operator<<(std::ostream& os, const json& j) { return os << j.dump(); }
	friend std::istream& operator>>(std::istream& is, json& j) {
		std::string text((std::istreambuf_iterator<char>(is)), std::istreambuf_iterator<char>());
		j = parse(text);
		return is;
	}
	friend bool operator==(const json& x, const json& y) {
		if (x.is_number() && y.is_number()) return x.number() == y.number();
		if (x.t != y.t) return false;
		switch (x.t) {
		case value_t::null: return true;
		case value_t::boolean: return x.b == y.b;
		case value_t::string: case value_t::binary: return x.s == y.s;
		case value_t::array: return x.a == y.a;
		case value_t::object: return x.o == y.o;
		default: return false;
		}
	}
	friend bool operator!=(const json& x, const json& y) { return !(x == y); }
	template <class C> friend bool operator==(const json& x, C* str) { return x.t == value_t::string && x.s == str; }
	template <class C> friend bool operator!=(const json& x, C* str) { return !(x == str); }

private:
	value_t t = value_t::null;
	bool b = false;
	int64_t i = 0;
	uint64_t u = 0;
	double f = 0.0;
	std::string s;
	array_t a;
	object_t o;

	static void be(std::string& out, uint64_t v, int n) { for (int k = n - 1; k >= 0; --k) out += (char)((v >> (8 * k)) & 0xff); }
	static void msgpack_uint(std::string& out, uint64_t v) {
		if (v <= 0x7f) out += (char)v;
		else if (v <= 0xff) { out += (char)0xcc; be(out, v, 1); }
		else if (v <= 0xffff) { out += (char)0xcd; be(out, v, 2); }
		else if (v <= 0xffffffffull) { out += (char)0xce; be(out, v, 4); }
		else { out += (char)0xcf; be(out, v, 8); }
	}
	void msgpack_to(std::string& out) const {
		switch (t) {
		case value_t::null: out += (char)0xc0; break;
		case value_t::boolean: out += (char)(b ? 0xc3 : 0xc2); break;
		case value_t::number_unsigned: msgpack_uint(out, u); break;
		case value_t::number_integer:
			if (i >= 0) msgpack_uint(out, (uint64_t)i);
			else if (i >= -32) out += (char)(int8_t)i;
			else if (i >= -128) { out += (char)0xd0; be(out, (uint64_t)i, 1); }
			else if (i >= -32768) { out += (char)0xd1; be(out, (uint64_t)i, 2); }
			else if (i >= -2147483648ll) { out += (char)0xd2; be(out, (uint64_t)i, 4); }
			else { out += (char)0xd3; be(out, (uint64_t)i, 8); }
			break;
		case value_t::number_float: {
			const float as_float = (float)f;
			if ((double)as_float == f) { uint32_t w; memcpy(&w, &as_float, 4); out += (char)0xca; be(out, w, 4); }
			else { uint64_t w; memcpy(&w, &f, 8); out += (char)0xcb; be(out, w, 8); }
			break;
		}
		case value_t::string:
			if (s.size() <= 31) out += (char)(0xa0 | s.size());
			else if (s.size() <= 0xff) { out += (char)0xd9; be(out, s.size(), 1); }
			else if (s.size() <= 0xffff) { out += (char)0xda; be(out, s.size(), 2); }
			else { out += (char)0xdb; be(out, s.size(), 4); }
			out += s;
			break;
		case value_t::binary:
			if (s.size() <= 0xff) { out += (char)0xc4; be(out, s.size(), 1); }
			else if (s.size() <= 0xffff) { out += (char)0xc5; be(out, s.size(), 2); }
			else { out += (char)0xc6; be(out, s.size(), 4); }
			out += s;
			break;
		case value_t::array:
			if (a.size() <= 15) out += (char)(0x90 | a.size());
			else if (a.size() <= 0xffff) { out += (char)0xdc; be(out, a.size(), 2); }
			else { out += (char)0xdd; be(out, a.size(), 4); }
			for (const json& e : a) e.msgpack_to(out);
			break;
		case value_t::object:
			if (o.size() <= 15) out += (char)(0x80 | o.size());
			else if (o.size() <= 0xffff) { out += (char)0xde; be(out, o.size(), 2); }
			else { out += (char)0xdf; be(out, o.size(), 4); }
			for (const auto& kv : o) { json(kv.first).msgpack_to(out); kv.second.msgpack_to(out); }
			break;
		}
	}
	static void dump_string(const std::string& v, std::string& out) {
		out += '"';
		for (unsigned char c : v) {
			switch (c) {
			case '"': out += "\\\""; break;
			case '\\': out += "\\\\"; break;
			case '\n': out += "\\n"; break;
			case '\r': out += "\\r"; break;
			case '\t': out += "\\t"; break;
			case '\b': out += "\\b"; break;
			case '\f': out += "\\f"; break;
			default:
				if (c < 0x20) { char buf[8]; snprintf(buf, sizeof(buf), "\\u%04x", c); out += buf; }
				else out += (char)c;
			}
		}
		out += '"';
	}
	void dump_to(std::string& out) const {
		char buf[40];
		switch (t) {
		case value_t::null: out += "null"; break;
		case value_t::boolean: out += b ? "true" : "false"; break;
		case value_t::number_integer: snprintf(buf, sizeof(buf), "%lld", (long long)i); out += buf; break;
		case value_t::number_unsigned: snprintf(buf, sizeof(buf), "%llu", (unsigned long long)u); out += buf; break;
		case value_t::number_float:
			if (!std::isfinite(f)) { out += "null"; break; } // nlohmann: NaN / infinity are serialised as null
			snprintf(buf, sizeof(buf), "%.17g", f);
			out += buf;
			if (!strpbrk_any(buf)) out += ".0"; // nlohmann keeps a float recognisable as one ("1.0", not "1")
			break;
		case value_t::string: dump_string(s, out); break;
		case value_t::binary: out += "{\"bytes\":["; for (size_t k = 0; k < s.size(); ++k) { if (k) out += ','; out += std::to_string((unsigned)(unsigned char)s[k]); } out += "],\"subtype\":null}"; break;
		case value_t::array:
			out += '[';
			for (size_t k = 0; k < a.size(); ++k) { if (k) out += ','; a[k].dump_to(out); }
			out += ']';
			break;
		case value_t::object:
			out += '{';
			for (size_t k = 0; k < o.size(); ++k) { if (k) out += ','; dump_string(o[k].first, out); out += ':'; o[k].second.dump_to(out); }
			out += '}';
			break;
		}
	}
	static bool strpbrk_any(const char* sbuf) { for (const char* q = sbuf; *q; ++q) if (*q == '.' || *q == 'e' || *q == 'E' || *q == 'n' || *q == 'i') return true; return false; }
	static void skip_ws(const std::string& tx, size_t& p) { while (p < tx.size() && (tx[p] == ' ' || tx[p] == '\t' || tx[p] == '\n' || tx[p] == '\r')) ++p; }
	static std::string parse_string(const std::string& tx, size_t& p) {
		std::string out;
		++p; // opening quote
		for (;;) {
			if (p >= tx.size()) throw std::runtime_error("json: unterminated string");
			const char c = tx[p++];
			if (c == '"') break;
			if (c != '\\') { out += c; continue; }
			if (p >= tx.size()) throw std::runtime_error("json: unterminated escape");
			const char e = tx[p++];
			switch (e) {
			case '"': out += '"'; break; case '\\': out += '\\'; break; case '/': out += '/'; break;
			case 'n': out += '\n'; break; case 'r': out += '\r'; break; case 't': out += '\t'; break; case 'b': out += '\b'; break; case 'f': out += '\f'; break;
			case 'u': {
				if (p + 4 > tx.size()) throw std::runtime_error("json: bad \\u escape");
				const unsigned cp = (unsigned)strtoul(tx.substr(p, 4).c_str(), nullptr, 16);
				p += 4;
				if (cp < 0x80) out += (char)cp;
				else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
				else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
				break;
			}
			default: throw std::runtime_error("json: bad escape");
			}
		}
		return out;
	}
	static json parse_value(const std::string& tx, size_t& p, int depth) {
		if (depth > 200) throw std::runtime_error("json: nesting too deep");
		skip_ws(tx, p);
		if (p >= tx.size()) throw std::runtime_error("json: unexpected end of input");
		json j;
		const char c = tx[p];
		if (c == '{') {
			j.t = value_t::object;
			++p; skip_ws(tx, p);
			if (p < tx.size() && tx[p] == '}') { ++p; return j; }
			for (;;) {
				skip_ws(tx, p);
				if (p >= tx.size() || tx[p] != '"') throw std::runtime_error("json: expected a key");
				const std::string key = parse_string(tx, p);
				skip_ws(tx, p);
				if (p >= tx.size() || tx[p] != ':') throw std::runtime_error("json: expected ':'");
				++p;
				j[key] = parse_value(tx, p, depth + 1);
				skip_ws(tx, p);
				if (p < tx.size() && tx[p] == ',') { ++p; continue; }
				if (p < tx.size() && tx[p] == '}') { ++p; break; }
				throw std::runtime_error("json: expected ',' or '}'");
			}
		} else if (c == '[') {
			j.t = value_t::array;
			++p; skip_ws(tx, p);
			if (p < tx.size() && tx[p] == ']') { ++p; return j; }
			for (;;) {
				j.a.push_back(parse_value(tx, p, depth + 1));
				skip_ws(tx, p);
				if (p < tx.size() && tx[p] == ',') { ++p; continue; }
				if (p < tx.size() && tx[p] == ']') { ++p; break; }
				throw std::runtime_error("json: expected ',' or ']'");
			}
		} else if (c == '"') {
			j.t = value_t::string;
			j.s = parse_string(tx, p);
		} else if (tx.compare(p, 4, "true") == 0) { j.t = value_t::boolean; j.b = true; p += 4; }
		else if (tx.compare(p, 5, "false") == 0) { j.t = value_t::boolean; j.b = false; p += 5; }
		else if (tx.compare(p, 4, "null") == 0) { p += 4; }
		else {
			size_t q = p;
			bool is_float = false;
			if (q < tx.size() && tx[q] == '-') ++q;
			while (q < tx.size() && ((tx[q] >= '0' && tx[q] <= '9') || tx[q] == '.' || tx[q] == 'e' || tx[q] == 'E' || tx[q] == '+' || tx[q] == '-')) {
				if (tx[q] == '.' || tx[q] == 'e' || tx[q] == 'E') is_float = true;
				++q;
			}
			if (q == p) throw std::runtime_error("json: unexpected character");
			const std::string num = tx.substr(p, q - p);
			if (is_float) { j.t = value_t::number_float; j.f = strtod(num.c_str(), nullptr); }
			else if (num[0] == '-') { j.t = value_t::number_integer; j.i = strtoll(num.c_str(), nullptr, 10); }
			else { j.t = value_t::number_unsigned; j.u = strtoull(num.c_str(), nullptr, 10); }
			p = q;
		}
		return j;
	}
};

namespace detail {
// ---- built-in conversions (nlohmann::detail::to_json / from_json) ----
inline void to_json(json& j, bool v) { j.set_bool(v); }
template <class T, typename std::enable_if<std::is_integral<T>::value && std::is_signed<T>::value && !std::is_same<T, bool>::value, int>::type = 0>
inline void to_json(json& j, T v) { j.set_int((int64_t)v); }
template <class T, typename std::enable_if<std::is_integral<T>::value && std::is_unsigned<T>::value && !std::is_same<T, bool>::value, int>::type = 0>
inline void to_json(json& j, T v) { j.set_uint((uint64_t)v); }
template <class T, typename std::enable_if<std::is_floating_point<T>::value, int>::type = 0>
inline void to_json(json& j, T v) { j.set_float((double)v); }
template <class T, typename std::enable_if<std::is_enum<T>::value, int>::type = 0>
inline void to_json(json& j, T v) { j.set_int((int64_t)v); }
inline void to_json(json& j, const std::string& v) { j.set_string(v); }
inline void to_json(json& j, const char* v) { j.set_string(v); }
template <class T> inline void to_json(json& j, const std::vector<T>& v) {
	j.set_array();
	for (const auto& e : v) j.push_back(json(static_cast<const T&>(e)));
}
inline void to_json(json& j, const std::vector<bool>& v) {
	j.set_array();
	for (bool e : v) j.push_back(json(e));
}
template <class T> inline void to_json(json& j, const std::map<std::string, T>& v) {
	j.set_object();
	for (const auto& kv : v) j[kv.first] = json(kv.second);
}

inline void from_json(const json& j, bool& v) {
	if (!j.is_boolean()) throw std::runtime_error("json: type must be boolean");
	v = j.raw_bool();
}
template <class T, typename std::enable_if<std::is_arithmetic<T>::value && !std::is_same<T, bool>::value, int>::type = 0>
inline void from_json(const json& j, T& v) {
	switch (j.type()) {
	case json::value_t::number_integer: v = (T)j.raw_int(); break;
	case json::value_t::number_unsigned: v = (T)j.raw_uint(); break;
	case json::value_t::number_float: v = (T)j.raw_float(); break;
	case json::value_t::boolean: v = (T)j.raw_bool(); break;
	default: throw std::runtime_error("json: type must be number");
	}
}
template <class T, typename std::enable_if<std::is_enum<T>::value, int>::type = 0>
inline void from_json(const json& j, T& v) { v = (T)(int64_t)j.number(); }
inline void from_json(const json& j, std::string& v) {
	if (!j.is_string()) throw std::runtime_error("json: type must be string");
	v = j.raw_string();
}
template <class T> inline void from_json(const json& j, std::vector<T>& v) {
	if (!j.is_array()) throw std::runtime_error("json: type must be array");
	v.clear();
	for (const json& e : j) v.push_back(e.template get<T>());
}

template <class T> void to_json_fn::operator()(json& j, T&& v) const { to_json(j, std::forward<T>(v)); }   // unqualified: built-ins here + ADL on T
template <class T> void from_json_fn::operator()(const json& j, T& v) const { from_json(j, v); }
} // namespace detail

} // namespace nlohmann
