// Minimal CSV row reader for the per-scene ".params" files (role of Include/CsvParser.hpp / Source/CsvParser.cpp).
#pragma once
#include <istream>
#include <sstream>
#include <string>
#include <vector>

class CSVRow
{
public:
	const std::string& operator[](size_t index) const { return mData.at(index); }
	size_t size() const { return mData.size(); }
	bool readNextRow(std::istream& str)
	{
		std::string line;
		mData.clear();
		while (std::getline(str, line)) {
			if (line.find_first_not_of(" \t\r\n") == std::string::npos) continue;
			std::stringstream ls(line);
			std::string cell;
			while (std::getline(ls, cell, ',')) mData.push_back(cell);
			return true;
		}
		return false;
	}
private:
	std::vector<std::string> mData;
};
