// json.h -- minimal JSON reader for the scene description (own code; the reference links json11,
// which is not vendored: scene.cpp:12-13).  Mirrors the json11 access idiom the reference relies
// on: operator[] on a missing key / wrong type yields a null value whose number_value() is 0,
// string_value() is "" and array_items() is empty (scene.cpp:57-250 depend on exactly that).
#pragma once
#include <cstdlib>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace glrt {

class Json {
public:
    enum Type { NUL, NUMBER, BOOL, STRING, ARRAY, OBJECT };
    Json() = default;

    static Json parse(const std::string &text, std::string &err) {
        Parser p{text, 0, err};
        Json v = p.value();
        p.ws();
        if (err.empty() && p.i != text.size()) err = "trailing characters after JSON value";
        return err.empty() ? v : Json();
    }

    bool is_null() const { return type_ == NUL; }
    bool is_number() const { return type_ == NUMBER; }
    bool is_string() const { return type_ == STRING; }
    bool is_array() const { return type_ == ARRAY; }
    bool is_object() const { return type_ == OBJECT; }
    double number_value() const { return type_ == NUMBER ? num_ : 0.0; }
    int int_value() const { return (int)number_value(); }
    bool bool_value() const { return type_ == BOOL && num_ != 0.0; }
    const std::string &string_value() const { static const std::string e; return type_ == STRING ? str_ : e; }
    const std::vector<Json> &array_items() const { static const std::vector<Json> e; return type_ == ARRAY ? *arr_ : e; }
    const Json &operator[](size_t i) const { return (type_ == ARRAY && i < arr_->size()) ? (*arr_)[i] : null(); }
    const Json &operator[](const std::string &k) const {
        if (type_ != OBJECT) return null();
        auto it = obj_->find(k);
        return it == obj_->end() ? null() : it->second;
    }

private:
    static const Json &null() { static const Json n; return n; }
    Type type_ = NUL;
    double num_ = 0.0;
    std::string str_;
    std::shared_ptr<std::vector<Json>> arr_;
    std::shared_ptr<std::map<std::string, Json>> obj_;

    struct Parser {
        const std::string &s;
        size_t i;
        std::string &err;
        void fail(const char *m) { if (err.empty()) err = std::string(m) + " at offset " + std::to_string(i); }
        void ws() {
            for (;;) {
                while (i < s.size() && (s[i] == ' ' || s[i] == '\t' || s[i] == '\n' || s[i] == '\r')) i++;
                if (i + 1 < s.size() && s[i] == '/' && s[i + 1] == '/') { while (i < s.size() && s[i] != '\n') i++; continue; }
                if (i + 1 < s.size() && s[i] == '/' && s[i + 1] == '*') {
                    i += 2;
                    while (i + 1 < s.size() && !(s[i] == '*' && s[i + 1] == '/')) i++;
                    i = i + 2 <= s.size() ? i + 2 : s.size();
                    continue;
                }
                break;
            }
        }
        Json value() {
            ws();
            if (i >= s.size()) { fail("unexpected end of input"); return Json(); }
            const char c = s[i];
            if (c == '{') return object();
            if (c == '[') return array();
            if (c == '"') { Json j; j.type_ = STRING; j.str_ = string(); return j; }
            if (s.compare(i, 4, "true") == 0) { i += 4; Json j; j.type_ = BOOL; j.num_ = 1; return j; }
            if (s.compare(i, 5, "false") == 0) { i += 5; Json j; j.type_ = BOOL; return j; }
            if (s.compare(i, 4, "null") == 0) { i += 4; return Json(); }
            char *end = nullptr;
            const double d = std::strtod(s.c_str() + i, &end);
            if (end == s.c_str() + i) { fail("unexpected character"); return Json(); }
            i = (size_t)(end - s.c_str());
            Json j; j.type_ = NUMBER; j.num_ = d;
            return j;
        }
        std::string string() {
            std::string out;
            i++;  // opening quote
            while (i < s.size() && s[i] != '"') {
                char c = s[i++];
                if (c == '\\' && i < s.size()) {
                    const char e = s[i++];
                    switch (e) {
                        case 'n': c = '\n'; break; case 't': c = '\t'; break; case 'r': c = '\r'; break;
                        case 'b': c = '\b'; break; case 'f': c = '\f'; break;
                        case 'u': {  // \uXXXX: keep ASCII, replace others with '?'
                            unsigned v = 0;
                            for (int k = 0; k < 4 && i < s.size(); k++) v = v * 16 + (unsigned)std::strtol(std::string(1, s[i++]).c_str(), nullptr, 16);
                            c = v < 128 ? (char)v : '?';
                            break;
                        }
                        default: c = e;
                    }
                }
                out.push_back(c);
            }
            if (i >= s.size()) fail("unterminated string"); else i++;
            return out;
        }
        Json array() {
            Json j; j.type_ = ARRAY; j.arr_ = std::make_shared<std::vector<Json>>();
            i++;
            ws();
            if (i < s.size() && s[i] == ']') { i++; return j; }
            for (;;) {
                j.arr_->push_back(value());
                if (!err.empty()) return Json();
                ws();
                if (i < s.size() && s[i] == ',') { i++; continue; }
                if (i < s.size() && s[i] == ']') { i++; return j; }
                fail("expected ',' or ']'");
                return Json();
            }
        }
        Json object() {
            Json j; j.type_ = OBJECT; j.obj_ = std::make_shared<std::map<std::string, Json>>();
            i++;
            ws();
            if (i < s.size() && s[i] == '}') { i++; return j; }
            for (;;) {
                ws();
                if (i >= s.size() || s[i] != '"') { fail("expected string key"); return Json(); }
                std::string k = string();
                ws();
                if (i >= s.size() || s[i] != ':') { fail("expected ':'"); return Json(); }
                i++;
                (*j.obj_)[k] = value();
                if (!err.empty()) return Json();
                ws();
                if (i < s.size() && s[i] == ',') { i++; continue; }
                if (i < s.size() && s[i] == '}') { i++; return j; }
                fail("expected ',' or '}'");
                return Json();
            }
        }
    };
};

}  // namespace glrt
