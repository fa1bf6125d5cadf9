// args.h — tiny option parser with the option-string grammar the reference hands to gclib's GArgs
// ("long;long;...;SMh" + "o:" for short options that take a value), enough to keep the tiebrush /
// tiecov command lines (tiebrush.cpp:605, tiecov.cpp:533) working verbatim.
#pragma once
#include <map>
#include <string>
#include <vector>

class Args {
 public:
  Args(int argc, char** argv, const char* fmt);
  // value of a short (single char) or long option; nullptr when absent; "" for flags that are present
  const char* getOpt(const char* name) const;
  const char* getOpt(char c) const {
    char b[2] = {c, 0};
    return getOpt(b);
  }
  int startNonOpt() {
    pos_ = 0;
    return (int)nonopt_.size();
  }
  const char* nextNonOpt() { return pos_ < nonopt_.size() ? nonopt_[pos_++].c_str() : nullptr; }
  const std::string& error() const { return err_; }
  void printCmdLine(FILE* f) const;

 private:
  std::map<std::string, std::string> opts_;
  std::vector<std::string> nonopt_, argv_;
  size_t pos_ = 0;
  std::string err_;
};
