// Buffered tokenizer over a memory-mapped text file, shared by the scene-text parser and the OBJ reader.
//
// The reference reads its scene files with `std::istream >> char / float / int` (reference src/main_cli.cpp:99-141):
// whitespace-separated tokens, one non-blank character per tag, and -- once a number fails to parse -- a failed
// stream, which ends the parse loop.  TextCursor keeps those observable rules on a plain
// [begin, end) byte range with std::from_chars, which is what makes 10^6 'T' lines load in a fraction of a second:
//   tag()      next non-blank character; false at the end of the input or after a failed number
//   f32()/i32  next number, leading '+' accepted like the stream's num_get; a token that is not a number
//              (or does not fit) fails the cursor, after which every read returns false
//   line ops   for the line-oriented OBJ format: rest_of_line(), at_line_end()
#pragma once
#include <charconv>
#include <cstddef>
#include <cstdint>
#include <string>
#include <string_view>
#include <system_error>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace hpt_host {

// read-only mapping of a whole file (empty files map to an empty range)
class MappedFile {
public:
    MappedFile() = default;
    MappedFile(const MappedFile &) = delete;
    MappedFile &operator=(const MappedFile &) = delete;
    ~MappedFile(){ close(); }
    bool open(const std::string &path){
        close();
        int fd = ::open(path.c_str(), O_RDONLY);
        if(fd < 0) return false;
        struct stat st;
        if(fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)){ ::close(fd); return false; }
        size_ = (size_t) st.st_size;
        if(size_ > 0){
            void *p = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd, 0);
            if(p == MAP_FAILED){ ::close(fd); size_ = 0; return false; }
            madvise(p, size_, MADV_SEQUENTIAL);
            data_ = (const char *) p;
        }
        ::close(fd);
        opened_ = true;
        return true;
    }
    void close(){
        if(data_) munmap((void *) data_, size_);
        data_ = nullptr; size_ = 0; opened_ = false;
    }
    const char *begin() const { return data_; }
    const char *end() const { return data_ + size_; }
    size_t size() const { return size_; }
    bool is_open() const { return opened_; }
private:
    const char *data_ = nullptr;
    size_t size_ = 0;
    bool opened_ = false;
};

class TextCursor {
public:
    TextCursor(const char *b, const char *e) : p_(b), end_(e) {}
    bool good() const { return good_; }

    bool tag(char &c){
        if(!good_) return false;
        skip_blank();
        if(p_ >= end_){ good_ = false; return false; }
        c = *p_++;
        return true;
    }
    bool f32(float &v){ return number(v); }
    bool i32(int &v){ return number(v); }
    template <size_t N> bool f32s(float (&v)[N]){ for(size_t i = 0; i < N; ++i) if(!number(v[i])) return false; return true; }
    bool f32s(float *v, int n){ for(int i = 0; i < n; ++i) if(!number(v[i])) return false; return true; }

    // everything up to and including the next '\n' (the stream's getline)
    void skip_line(){
        while(p_ < end_ && *p_ != '\n') ++p_;
        if(p_ < end_) ++p_;
    }
    // ---- line-oriented helpers (OBJ) ----
    void skip_inline_blank(){ while(p_ < end_ && (*p_ == ' ' || *p_ == '\t' || *p_ == '\r')) ++p_; }
    bool at_line_end(){ skip_inline_blank(); return p_ >= end_ || *p_ == '\n'; }
    bool at_end() const { return p_ >= end_; }
    // next blank-delimited word on the current line ("" at the end of the line)
    std::string_view word(){
        skip_inline_blank();
        const char *b = p_;
        while(p_ < end_ && !is_blank(*p_)) ++p_;
        return std::string_view(b, (size_t) (p_ - b));
    }
    // drops the rest of the current word (e.g. the "/7/3" behind a face's vertex index)
    void skip_word(){ while(p_ < end_ && !is_blank(*p_)) ++p_; }
    // a number that must sit on the current line; does not fail the cursor
    template <typename T> bool number_on_line(T &v){
        skip_inline_blank();
        if(p_ >= end_ || *p_ == '\n') return false;
        const char *q = p_;
        if(*q == '+') ++q;
        auto r = std::from_chars(q, end_, v);
        if(r.ec != std::errc() || r.ptr == q) return false;
        p_ = r.ptr;
        return true;
    }

private:
    static bool is_blank(char c){ return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }
    void skip_blank(){ while(p_ < end_ && is_blank(*p_)) ++p_; }

    template <typename T> bool number(T &v){
        if(!good_) return false;
        skip_blank();
        const char *q = p_;
        if(q < end_ && *q == '+') ++q;
        // the stream's num_get only starts on a sign, a digit or (for floats) a '.'; from_chars would also take "inf"/"nan"
        const char *d = (q < end_ && *q == '-') ? q + 1 : q;
        bool starts = d < end_ && ((*d >= '0' && *d <= '9') || (std::is_floating_point<T>::value && *d == '.'));
        if(!starts || (q != p_ && q < end_ && *q == '-')){ good_ = false; v = T(0); return false; }
        auto r = std::from_chars(q, end_, v);
        if(r.ec != std::errc()){ good_ = false; v = T(0); return false; }
        p_ = r.ptr;
        return true;
    }

    const char *p_, *end_;
    bool good_ = true;
};

} // namespace hpt_host
