// dvo_benchmark/file_reader.h -- line-oriented reader of whitespace-separated record files (assoc.txt, groundtruth.txt).
// Same public interface as the reference's dvo_benchmark::FileReader<EntryT>
// (dvo_benchmark/include/dvo_benchmark/file_reader.h:34-113): EntryT must support `std::istream >> EntryT`.
#pragma once

#include <fstream>
#include <limits>
#include <string>
#include <vector>

namespace dvo_benchmark {

template <class EntryT>
class FileReader {
 public:
  explicit FileReader(const std::string& file) : has_entry_(false), file_(file), stream_(file.c_str()) {}
  virtual ~FileReader() {}

  bool good() const { return stream_.good(); }

  // drop `num_lines` lines
  void skip(int num_lines) {
    for (int i = 0; i < num_lines && stream_.good(); ++i) stream_.ignore(std::numeric_limits<std::streamsize>::max(), '\n');
  }

  // drop the leading '#' lines (the TUM files start with a three-line comment header)
  void skipComments() {
    while (stream_.good() && stream_.peek() == '#') skip(1);
  }

  // advance to the next record; false once the file is exhausted
  bool next() {
    if (!stream_.good() || stream_.eof()) return false;
    EntryT e;
    stream_ >> e;
    if (stream_.fail()) return false;     // trailing newline / malformed tail: keep the last good entry
    entry_ = e;
    has_entry_ = true;
    return true;
  }

  // the current record and everything after it
  void readAllEntries(std::vector<EntryT>& entries) {
    if (!has_entry_ && !next()) return;
    do entries.push_back(entry_);
    while (next());
  }

  const EntryT& entry() const { return entry_; }
  EntryT& entry() { return entry_; }
  bool hasEntry() const { return has_entry_; }

 private:
  EntryT entry_;
  bool has_entry_;
  std::string file_;
  std::ifstream stream_;
};

}  // namespace dvo_benchmark
