// Minimal JSON reader (objects, arrays, strings, numbers, true/false/null) for the glTF loader.  No dependencies.
#pragma once
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace gmupt {

struct Json
{
	enum Type { Null, Bool, Number, String, Array, Object } type = Null;
	bool b = false;
	double num = 0.0;
	std::string str;
	std::vector<Json> arr;
	std::map<std::string, Json> obj;

	bool has(const std::string& k) const { return type == Object && obj.count(k) != 0; }
	const Json& operator[](const std::string& k) const
	{
		static const Json null;
		if (type != Object) return null;
		auto it = obj.find(k);
		return it == obj.end() ? null : it->second;
	}
	const Json& operator[](size_t i) const { static const Json null; return (type == Array && i < arr.size()) ? arr[i] : null; }
	size_t size() const { return type == Array ? arr.size() : type == Object ? obj.size() : 0; }
	double number(double def = 0.0) const { return type == Number ? num : def; }
	int integer(int def = -1) const { return type == Number ? static_cast<int>(num) : def; }
	const std::string& string() const { return str; }

	static Json parse(const std::string& text)
	{
		size_t pos = 0;
		Json v = parseValue(text, pos);
		skip(text, pos);
		if (pos != text.size()) throw std::runtime_error("JSON: trailing characters");
		return v;
	}

private:
	static void skip(const std::string& s, size_t& p) { while (p < s.size() && (s[p] == ' ' || s[p] == '\n' || s[p] == '\t' || s[p] == '\r')) p++; }
	static Json parseValue(const std::string& s, size_t& p)
	{
		skip(s, p);
		if (p >= s.size()) throw std::runtime_error("JSON: unexpected end");
		Json v;
		const char c = s[p];
		if (c == '{') {
			v.type = Object; p++; skip(s, p);
			if (p < s.size() && s[p] == '}') { p++; return v; }
			for (;;) {
				skip(s, p);
				Json key = parseString(s, p);
				skip(s, p);
				if (p >= s.size() || s[p] != ':') throw std::runtime_error("JSON: expected ':'");
				p++;
				v.obj[key.str] = parseValue(s, p);
				skip(s, p);
				if (p < s.size() && s[p] == ',') { p++; continue; }
				if (p < s.size() && s[p] == '}') { p++; break; }
				throw std::runtime_error("JSON: expected ',' or '}'");
			}
		} else if (c == '[') {
			v.type = Array; p++; skip(s, p);
			if (p < s.size() && s[p] == ']') { p++; return v; }
			for (;;) {
				v.arr.push_back(parseValue(s, p));
				skip(s, p);
				if (p < s.size() && s[p] == ',') { p++; continue; }
				if (p < s.size() && s[p] == ']') { p++; break; }
				throw std::runtime_error("JSON: expected ',' or ']'");
			}
		} else if (c == '"') {
			v = parseString(s, p);
		} else if (s.compare(p, 4, "true") == 0) { v.type = Bool; v.b = true; p += 4; }
		else if (s.compare(p, 5, "false") == 0) { v.type = Bool; v.b = false; p += 5; }
		else if (s.compare(p, 4, "null") == 0) { p += 4; }
		else {
			char* end = nullptr;
			v.num = std::strtod(s.c_str() + p, &end);
			if (end == s.c_str() + p) throw std::runtime_error("JSON: bad value");
			v.type = Number; p = static_cast<size_t>(end - s.c_str());
		}
		return v;
	}
	static Json parseString(const std::string& s, size_t& p)
	{
		if (p >= s.size() || s[p] != '"') throw std::runtime_error("JSON: expected string");
		Json v; v.type = String; p++;
		while (p < s.size() && s[p] != '"') {
			if (s[p] == '\\' && p + 1 < s.size()) {
				const char e = s[p + 1];
				if (e == 'n') v.str += '\n'; else if (e == 't') v.str += '\t'; else if (e == 'u') { v.str += '?'; p += 4; } else v.str += e;
				p += 2;
			} else v.str += s[p++];
		}
		if (p >= s.size()) throw std::runtime_error("JSON: unterminated string");
		p++;
		return v;
	}
};

} // namespace gmupt
